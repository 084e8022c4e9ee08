#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Build-container only: imports `/root/reference` (jinshisai/SLAM) with the inert
third-party stubs of `_refstubs.py`, drives `ChannelFit.makegrid`/`cubemodel`/
`fitting` on small seeded synthetic cubes and stores inputs + outputs as .npz.
No reference source is copied; the vectors are data.  The GPU box never runs
this script (it has no /root/reference) -- it only reads the .npz files.

Usage:  python tests/golden/make_golden.py          (rewrites tests/golden/*.npz)
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refstubs  # noqa: E402

_refstubs.install()
sys.path.insert(0, '/root/reference')
import matplotlib  # noqa: E402

matplotlib.use('Agg')
import channelfit._channelfit as cf  # noqa: E402
from channelfit import ChannelFit  # noqa: E402

KEYS = ['Mstar', 'Rc', 'cs', 'h1', 'h2', 'pI', 'Rin', 'Ienv',
        'xoff', 'yoff', 'voff', 'incloff', 'paoff']


def base_params(**kw):
    p = dict(Mstar=0.5, Rc=150., cs=0.4, h1=0.1, h2=0.3, pI=1., Rin=0.,
             Ienv=0., xoff=0., yoff=0., voff=0., incloff=0., paoff=0.)
    p.update(kw)
    return p


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def build(cfg, data):
    chan = ChannelFit(disk=cfg['disk'], envelope=cfg['envelope'],
                      scaling='uniform', progressbar=False)
    x, y, v = cfg['x'], cfg['y'], cfg['v']
    chan.x, chan.y, chan.v = x, y, v
    chan.dx, chan.dy, chan.dv = x[1] - x[0], y[1] - y[0], v[1] - v[0]
    chan.nx, chan.ny, chan.nv = len(x), len(y), len(v)
    chan.bmaj, chan.bmin, chan.bpa = cfg['bmaj'], cfg['bmin'], cfg['bpa']
    chan.sigma, chan.data = cfg['sigma'], data
    chan.read_cubefits = lambda *a, **k: None
    quiet(chan.makegrid, cubefits='synthetic', pa=cfg['pa'], incl=cfg['incl'],
          dist=1, vsys=0, rmax=cfg['rmax'], vlim=cfg['vlim'],
          sigma=cfg['sigma'], nlayer=cfg['nlayer'],
          gaussmargin=cfg['gaussmargin'], signmajor=cfg['signmajor'],
          signminor=cfg['signminor'])
    chan.free = {k: True for k in chan.paramkeys}
    return chan


def evaluate(chan, p, convolving=True):
    """Run the reference forward pass and tap its stages."""
    model = chan.cubemodel(**p, convolving=convolving)
    Iunif = chan.get_Iunif(p['Mstar'], p['Rc'], p['pI'], p['Ienv'], p['voff'])
    if convolving:
        Iconv = cf.convolve(Iunif, [chan.gaussbeam], mode='same')
    else:
        Iconv = chan.peaktounity(Iunif)
    Ireg = chan.rgi2d(p['xoff'], p['yoff'], Iconv)
    gf = np.sum(Ireg * chan.data_valid)
    ff = np.sum(Ireg * Ireg)
    scale = chan.get_scale(Ireg)[0]
    chi2 = np.nansum((chan.data_valid - model)**2) / chan.sigma**2 \
        / chan.pixperbeam
    return dict(model=model, Iunif=Iunif, Iconv=Iconv, Ireg=Ireg, gf=gf, ff=ff,
                scale=scale, chi2=chi2, prof=chan.prof.copy(),
                prof_d=chan.prof_d, xdisk=chan.xdisk, vlos=chan.vlos)


def make_cube(cfg, seed, truth=None, snr=50.):
    """noise (+ truth model scaled to snr*sigma at peak), default_rng(seed)."""
    rng = np.random.default_rng(seed)
    shape = (len(cfg['v']), len(cfg['y']), len(cfg['x']))
    d = rng.normal(0, cfg['sigma'], shape)
    if truth is not None:
        d = d + truth
    return d


def grid_cfg(nx, ny, nv, dpix, dyfac=1.0, dv=0.25, vdesc=False, **kw):
    x = -(np.arange(nx) - nx // 2) * dpix            # descending, like R.A.
    y = (np.arange(ny) - ny // 2) * dpix * dyfac
    v = (np.arange(nv) - nv / 2 + 0.5) * dv
    if vdesc:
        v = v[::-1].copy()
    cfg = dict(x=x, y=y, v=v, sigma=1.7e-3, pa=2., incl=60., rmax=150.,
               vlim=(-100, 0, 0, 100), nlayer=3, gaussmargin=1.6,
               signmajor=1, signminor=1, disk=True, envelope=False,
               bmaj=5 * dpix, bmin=3.5 * dpix, bpa=0.)
    cfg.update(kw)
    return cfg


# ----------------------------------------------------------------------------
# Configurations and the parameter vectors each is evaluated on.
# 'full' vectors store Iunif / Iconv / model arrays; the rest store scalars
# and per-channel sums only (keeps the fixtures small).
# ----------------------------------------------------------------------------
def configs():
    out = {}
    # A: generic separable beam, 32x32x16, all channels valid.
    out['A'] = (grid_cfg(32, 32, 16, 10.),
                [('generic', base_params(), True),
                 ('thin_h1', base_params(h1=0.005), True),
                 ('no_h2', base_params(h2=-1.), True),
                 ('pI0', base_params(pI=0.), False),
                 ('pI_neg', base_params(pI=-0.7), False),
                 ('Rin', base_params(Rin=40.), True),
                 ('voff', base_params(voff=0.11), False),
                 ('xyoff', base_params(xoff=7.3, yoff=-4.1), True),
                 ('xyoff_big', base_params(xoff=95., yoff=-60.), True),
                 ('paoff', base_params(paoff=12.), True),
                 ('incloff', base_params(incloff=-8.), False),
                 ('cs_narrow', base_params(cs=0.001), True),
                 ('cs_mid', base_params(cs=0.075), False),
                 ('cs_wide', base_params(cs=1.0), False),
                 ('Mstar_hi', base_params(Mstar=2.3, Rc=60.), False),
                 ('smallRc', base_params(Rc=35., h1=0.0, h2=0.45), False),
                 ('flat_both', base_params(h1=0., h2=0.), False),
                 ('all_off', base_params(Mstar=0.8, Rc=110., cs=0.3, h1=0.2,
                                         h2=0.05, pI=1.7, Rin=12., xoff=-3.,
                                         yoff=2., voff=-0.07, incloff=5.,
                                         paoff=-9.), True)])
    # B: rotated beam, nx != ny, dy != dx, velocity windows with a gap, near
    #    edge-on so that |tan^-2 i - h^2| < 1e-3 selects the one-root branch.
    out['B'] = (grid_cfg(48, 40, 20, 8., dyfac=1.25, incl=85., bpa=20.,
                         bmaj=48., bmin=30., vlim=(-2.4, -0.3, 0.3, 2.4),
                         rmax=140.),
                [('generic', base_params(), True),
                 ('one_root', base_params(h1=0.0875, h2=-1.), True),
                 ('one_root_both', base_params(h1=0.0875, h2=0.0874), False),
                 ('Dneg', base_params(h1=0.4, h2=0.5, incloff=-20.), True),
                 ('xyoff', base_params(xoff=-11., yoff=6.5, voff=0.05), False),
                 ('paoff', base_params(paoff=-30., incloff=3.), False)])
    # C: envelope switched on (disk + infalling envelope beyond Rc).
    out['C'] = (grid_cfg(32, 32, 16, 10., envelope=True, incl=70.),
                [('env', base_params(Rc=70., Ienv=0.5), True),
                 ('env_Ienv0', base_params(Rc=70., Ienv=0.), False),
                 ('env_big', base_params(Rc=40., Ienv=3., pI=0.5, Rin=15.),
                  False)])
    # D: envelope only (disk=False).
    out['D'] = (grid_cfg(32, 32, 16, 10., envelope=True, disk=False, incl=70.),
                [('envonly', base_params(Rc=60., Ienv=1.), True)])
    # E: descending velocity axis (dv < 0 forces the boxcar profile), signs
    #    derived from the moment-1 map, nlayer = 2.
    out['E'] = (grid_cfg(32, 32, 16, 10., vdesc=True, signmajor=None,
                         signminor=None, nlayer=2),
                [('vdesc', base_params(), True)])
    # F: nlayer = 4 and a tighter Gaussian margin.
    out['F'] = (grid_cfg(32, 32, 16, 10., nlayer=4, gaussmargin=1.2),
                [('deep', base_params(), True)])
    return out


def record_config(name, cfg, vectors, outdir):
    # Truth cube = reference model for the generic parameters, scaled to
    # 50 sigma at peak, plus noise (seeded).  Signs fixed to +1 for the truth.
    cfg0 = dict(cfg)
    derive_signs = cfg['signmajor'] is None
    if derive_signs:
        cfg0['signmajor'], cfg0['signminor'] = -1, 1
    shape = (len(cfg['v']), len(cfg['y']), len(cfg['x']))
    chan0 = build(cfg0, np.zeros(shape))
    # data_valid is all-zero here, so cubemodel's scale is 0; take the unscaled
    # regridded cube as the truth shape instead.
    truth_valid = evaluate(chan0, vectors[0][1])['Ireg']
    truth = np.zeros(shape)
    sel = np.isin(cfg['v'], chan0.v_valid)
    order = [np.where(cfg['v'] == vv)[0][0] for vv in chan0.v_valid]
    truth[order] = truth_valid * (50 * cfg['sigma'] / truth_valid.max())
    assert sel.sum() == len(order)
    data = make_cube(cfg, seed=sum(map(ord, name)), truth=truth)

    chan = build(cfg, data)
    rec = {}
    for k in ('x', 'y', 'v'):
        rec['in_' + k] = cfg[k]
    rec['in_data'] = data
    for k in ('sigma', 'pa', 'incl', 'rmax', 'nlayer', 'gaussmargin', 'bmaj',
              'bmin', 'bpa'):
        rec['in_' + k] = np.float64(cfg[k])
    rec['in_vlim'] = np.array(cfg['vlim'], dtype=float)
    rec['in_disk'] = np.bool_(cfg['disk'])
    rec['in_envelope'] = np.bool_(cfg['envelope'])
    rec['in_signs_given'] = np.bool_(not derive_signs)
    # makegrid outputs (a0)
    for k in ('Xnest0', 'Ynest0', 'gaussbeam', 'xneed', 'yneed', 'v_valid',
              'data_valid'):
        rec['setup_' + k] = np.asarray(getattr(chan, k))
    for k in ('pixperbeam', 'signmajor', 'signminor', 'dv', 'incl0'):
        rec['setup_' + k] = np.float64(getattr(chan, k))
    for k in ('nq1', 'nq3', 'ineed0', 'ineed1', 'nlayer'):
        rec['setup_' + k] = np.int64(getattr(chan, k))
    rec['setup_npix'] = np.int64(chan.Xnest0.shape[-1])
    rec['setup_vunit'] = np.float64(cf.vunit)

    names = []
    for vname, p, full in vectors:
        r = evaluate(chan, p)
        names.append(vname)
        pre = f'vec_{vname}_'
        rec[pre + 'p'] = np.array([p[k] for k in KEYS])
        for k in ('gf', 'ff', 'scale', 'chi2', 'prof_d'):
            rec[pre + k] = np.float64(r[k])
        rec[pre + 'prof'] = r['prof']
        rec[pre + 'Iunif_chsum'] = r['Iunif'].sum(axis=(1, 2))
        rec[pre + 'Iconv_chsum'] = r['Iconv'].sum(axis=(1, 2))
        rec[pre + 'model_chsum'] = r['model'].sum(axis=(1, 2))
        if full:
            rec[pre + 'Iunif'] = r['Iunif']
            rec[pre + 'Iconv'] = r['Iconv']
            rec[pre + 'model'] = r['model']
    # geometry taps for the first vector
    r = evaluate(chan, vectors[0][1])
    for s in range(4):
        if r['xdisk'][s] is not None:
            rec[f'geom_xdisk{s}'] = r['xdisk'][s]
            rec[f'geom_vlos{s}'] = r['vlos'][s]
    # convolving=False tap (peaktounity) on the first vector
    r = evaluate(chan, vectors[0][1], convolving=False)
    rec['noconv_model'] = r['model']
    rec['noconv_chi2'] = np.float64(r['chi2'])
    rec['vector_names'] = np.array(names)
    np.savez_compressed(os.path.join(outdir, f'channelfit_{name}.npz'), **rec)
    print(f'config {name}: {len(names)} vectors, npix={int(rec["setup_npix"])}'
          f', nneed={len(chan.xneed)}, K={chan.gaussbeam.shape[0]}, '
          f'nv_valid={len(chan.v_valid)}')
    return chan, data


def record_fitting(cfg, data, outdir):
    """Capture what `fitting` hands to the sampler (:539-663): prior box,
    labels, the lnprob closure's values on seeded points, popt.txt layout."""
    captured = {}

    def fake_emcee_corner(plim, lnprob, simpleoutput=True, **kw):
        rng = np.random.default_rng(7)
        plim = np.array(plim)
        P = plim[0] + (plim[1] - plim[0]) * rng.random((6, plim.shape[1]))
        captured['plim'] = plim
        captured['labels'] = np.array(kw['labels'])
        captured['P'] = P
        captured['lnp'] = np.array([lnprob(p) for p in P])
        captured['kw_nwalkers_per_ndim'] = kw['nwalkers_per_ndim']
        return [P[0], P[1], P[2], P[3]]

    real = cf.emcee_corner
    cf.emcee_corner = fake_emcee_corner
    cwd = os.getcwd()
    try:
        chan = build(cfg, data)
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            fixed = {'cs': 0.4, 'Rin': 0, 'Ienv': 0, 'xoff': 0, 'yoff': 0,
                     'voff': 0, 'incloff': 0, 'paoff': 0}
            quiet(chan.fitting, Mstar_range=[0.1, 1.0], Rc_range=[30, 300],
                  pI_range=[0, 3], h1_range=[0, 0.5], h2_range=[0, 0.5],
                  fixed_params=fixed, filename='gold',
                  kwargs_emcee_corner={'nwalkers_per_ndim': 2, 'nburnin': 3,
                                       'nsteps': 3, 'rangelevel': 0.99})
            with open('gold.popt.txt') as f:
                popt_txt = f.read()
            # all-fixed branch (:628-643): reduced chi2
            chan2 = build(cfg, data)
            allfixed = dict(fixed, Mstar=0.5, Rc=150., h1=0.1, h2=0.3, pI=1.)
            quiet(chan2.fitting, fixed_params=allfixed, filename='gold2')
    finally:
        os.chdir(cwd)
        cf.emcee_corner = real
    rec = {'fit_' + k: v for k, v in captured.items()}
    rec['fit_popt_txt'] = np.array(popt_txt)
    rec['fit_popt'] = np.array([chan.popt[k] for k in KEYS], dtype=float)
    rec['fit_plow'] = np.array([chan.plow[k] for k in KEYS], dtype=float)
    rec['fit_chi2r_allfixed'] = np.float64(chan2.chi2r)
    rec['fit_pa_rad_after'] = np.float64(chan.pa_rad)
    np.savez_compressed(os.path.join(outdir, 'channelfit_fitting.npz'), **rec)
    print('fitting capture: lnp =', captured['lnp'])


def record_known_answer(outdir):
    """SURVEY.md Appendix A known answer (chi2 = 4229.974367626359)."""
    n, nv = 64, 32
    dpix = 400 / 64
    cfg = dict(x=-(np.arange(n) - 32) * dpix, y=(np.arange(n) - 32) * dpix,
               v=(np.arange(nv) - 16 + 0.5) * 0.25, sigma=1.7e-3, pa=2.,
               incl=85., rmax=190., vlim=(-4, -0.5, 0.5, 4), nlayer=3,
               gaussmargin=1.6, signmajor=1, signminor=1, disk=True,
               envelope=False, bmaj=6 * dpix, bmin=4 * dpix, bpa=20.)
    data = np.random.default_rng(0).normal(0, 1.7e-3, (nv, n, n))
    chan = build(cfg, data)
    r = evaluate(chan, base_params())
    assert r['chi2'] == 4229.974367626359, r['chi2']
    np.savez_compressed(os.path.join(outdir, 'channelfit_known.npz'),
                        chi2=np.float64(r['chi2']), gf=np.float64(r['gf']),
                        ff=np.float64(r['ff']),
                        model_chsum=r['model'].sum(axis=(1, 2)),
                        Iunif_chsum=r['Iunif'].sum(axis=(1, 2)))
    print('known answer reproduced:', r['chi2'])


def main():
    outdir = HERE
    keep = {}
    for name, (cfg, vectors) in configs().items():
        keep[name] = (cfg,) + record_config(name, cfg, vectors, outdir)
    cfgA, _, dataA = keep['A']
    record_fitting(cfgA, dataA, outdir)
    record_known_answer(outdir)


if __name__ == '__main__':
    main()
