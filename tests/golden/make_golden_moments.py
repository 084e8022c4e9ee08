#!/usr/bin/env python3
"""Golden vectors for the moment maps behind `plotmom`, made by
RUNNING THE REFERENCE (build container only, same stubs as make_golden.py).

For configurations A and B (cubes stored in channelfit_{A,B}.npz, scaling
'uniform') records the observed moment maps `makegrid` keeps, and per parameter
vector the maps `makemom01` (`_channelfit.py:40-50`) returns for the model cube,
which is what `plotmom('mod')` draws (`:724-733`).

Usage:  python tests/golden/make_golden_moments.py
"""
import os

import numpy as np

import make_golden_mom0 as mm      # installs the stubs, imports the reference
from make_golden import KEYS, cf

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rec = {}
    for cfgname in ('A', 'B'):
        f = np.load(os.path.join(HERE, f'channelfit_{cfgname}.npz'))
        chan = mm.build(f, 'uniform')
        rec[f'{cfgname}_obs_mom0'] = chan.mom0
        rec[f'{cfgname}_obs_mom1'] = chan.mom1
        rec[f'{cfgname}_sigma_mom0'] = np.float64(chan.sigma_mom0)
        names = []
        for vname, p in mm.VECTORS[cfgname][:3]:
            model = chan.cubemodel(**p)
            m = cf.makemom01(model, chan.v_valid, chan.sigma)
            pre = f'{cfgname}_vec_{vname}_'
            names.append(vname)
            rec[pre + 'p'] = np.array([p[k] for k in KEYS])
            rec[pre + 'mom0'] = m['mom0']
            rec[pre + 'mom1'] = m['mom1']
            print(cfgname, vname, 'mom0 sum', float(np.sum(m['mom0'])),
                  'mom1 finite', int(np.isfinite(m['mom1']).sum()), 'of', m['mom1'].size)
        rec[f'{cfgname}_vector_names'] = np.array(names)
    np.savez_compressed(os.path.join(HERE, 'channelfit_moments.npz'), **rec)


if __name__ == '__main__':
    main()
