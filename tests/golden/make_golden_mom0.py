#!/usr/bin/env python3
"""Golden vectors for the `mom0*` scalings of ChannelFit, made
by RUNNING THE REFERENCE (build container only, same stubs as make_golden.py).

Re-uses the data cubes of configurations A (separable beam) and B (rotated
beam, nx != ny, velocity gap) stored in channelfit_{A,B}.npz and records, for
scaling in {mom0ft, mom0clean, mom0model}: the deconvolved moment-0 map, the
x-flipped beam, and per parameter vector the regridded cube, the model cube and
chi2 (`_channelfit.py:343-371, 520-537, 601-602`).

Usage:  python tests/golden/make_golden_mom0.py
"""
import os

import numpy as np

import make_golden as mg          # installs the stubs, imports the reference
from make_golden import KEYS, base_params, cf, quiet

HERE = os.path.dirname(os.path.abspath(__file__))


def build(f, scaling):
    chan = cf.ChannelFit(disk=bool(f['in_disk']), envelope=bool(f['in_envelope']),
                         scaling=scaling, progressbar=False)
    x, y, v = f['in_x'], f['in_y'], f['in_v']
    chan.x, chan.y, chan.v = x, y, v
    chan.dx, chan.dy, chan.dv = x[1] - x[0], y[1] - y[0], v[1] - v[0]
    chan.nx, chan.ny, chan.nv = len(x), len(y), len(v)
    chan.bmaj, chan.bmin, chan.bpa = (float(f['in_bmaj']), float(f['in_bmin']),
                                      float(f['in_bpa']))
    chan.sigma, chan.data = float(f['in_sigma']), f['in_data']
    chan.read_cubefits = lambda *a, **k: None
    quiet(chan.makegrid, cubefits='synthetic', pa=float(f['in_pa']),
          incl=float(f['in_incl']), dist=1, vsys=0, rmax=float(f['in_rmax']),
          vlim=tuple(f['in_vlim']), sigma=float(f['in_sigma']),
          nlayer=int(f['in_nlayer']), gaussmargin=float(f['in_gaussmargin']),
          signmajor=int(f['setup_signmajor']), signminor=int(f['setup_signminor']))
    chan.free = {k: True for k in chan.paramkeys}
    return chan


VECTORS = {
    'A': [('generic', base_params()),
          ('xyoff', base_params(xoff=7.3, yoff=-4.1)),
          ('all_off', base_params(Mstar=0.8, Rc=110., cs=0.3, h1=0.2, h2=0.05,
                                  pI=1.7, Rin=12., xoff=-3., yoff=2., voff=-0.07,
                                  incloff=5., paoff=-9.)),
          ('xyoff_big', base_params(xoff=95., yoff=-60.))],
    'B': [('generic', base_params()),
          ('paoff', base_params(paoff=-30., incloff=3.))],
}


def main():
    for cfgname, scalings in (('A', ('mom0ft', 'mom0clean', 'mom0model')),
                              ('B', ('mom0ft', 'mom0clean'))):
        f = np.load(os.path.join(HERE, f'channelfit_{cfgname}.npz'))
        rec = {}
        for sc in scalings:
            chan = build(f, sc)
            pre = f'{sc}_'
            rec[pre + 'mom0decon'] = chan.mom0decon
            rec[pre + 'gaussbeam'] = chan.gaussbeam
            rec[pre + 'mom0'] = chan.mom0
            rec[pre + 'sigma_mom0'] = np.float64(chan.sigma_mom0)
            rec[pre + 'resdecon'] = chan.resdecon
            names = []
            for vi, (vname, p) in enumerate(VECTORS[cfgname]):
                model = chan.cubemodel(**p)
                chi2 = np.nansum((chan.data_valid - model)**2) / chan.sigma**2 \
                    / chan.pixperbeam
                vp = f'{pre}vec_{vname}_'
                names.append(vname)
                rec[vp + 'p'] = np.array([p[k] for k in KEYS])
                rec[vp + 'chi2'] = np.float64(chi2)
                rec[vp + 'model_chsum'] = model.sum(axis=(1, 2))
                if vi < 2:
                    rec[vp + 'model'] = model
                    Iu = chan.get_Iunif(p['Mstar'], p['Rc'], p['pI'], p['Ienv'], p['voff'])
                    rec[vp + 'Iregrid'] = chan.rgi2d(p['xoff'], p['yoff'], Iu)
                if vi == 0:
                    rec[vp + 'noconv_model'] = chan.cubemodel(**p, convolving=False)
            rec[pre + 'vector_names'] = np.array(names)
            print(cfgname, sc, 'mom0decon sum', float(np.sum(chan.mom0decon)),
                  'chi2', [float(rec[f'{pre}vec_{n}_chi2']) for n in names])
        rec['scalings'] = np.array(scalings)
        np.savez_compressed(os.path.join(HERE, f'channelfit_mom0_{cfgname}.npz'), **rec)


if __name__ == '__main__':
    main()
