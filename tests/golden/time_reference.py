#!/usr/bin/env python3
"""Time one likelihood evaluation of THE REFERENCE next to the oracle's, on the
same cube in the same process (build container only: imports /root/reference with
the stubs of make_golden.py).  Answers SURVEY.md 8d's question whether the oracle is
a fair stand-in for the reference as the CPU baseline of bench.py.

Usage:  python tests/golden/time_reference.py [nx nv]     (default 256 64 = config C2's shape)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402
from oracle import channelfit_oracle as orc  # noqa: E402


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nv = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    cfg = mg.grid_cfg(nx, nx, nv, 600. / nx, dv=6.4 / nv, rmax=250., nlayer=4)
    data = mg.make_cube(cfg, 7)
    p = mg.base_params()
    t0 = time.perf_counter()
    chan = mg.build(cfg, data)
    t_setup_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    s = orc.make_setup(x=cfg['x'], y=cfg['y'], v=cfg['v'], data=data, bmaj=cfg['bmaj'],
                       bmin=cfg['bmin'], bpa=cfg['bpa'], sigma=cfg['sigma'], pa=cfg['pa'],
                       incl=cfg['incl'], rmax=cfg['rmax'], vlim=cfg['vlim'], nlayer=cfg['nlayer'],
                       gaussmargin=cfg['gaussmargin'], signmajor=cfg['signmajor'],
                       signminor=cfg['signminor'], disk=cfg['disk'], envelope=cfg['envelope'])
    t_setup_orc = time.perf_counter() - t0
    pv = np.array([p[k] for k in mg.KEYS])

    def ref_eval():
        m = chan.cubemodel(**p)
        return np.nansum((chan.data_valid - m)**2) / chan.sigma**2 / chan.pixperbeam

    def best(fn, n=3):
        ts, out = [], None
        for _ in range(n):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        return min(ts), out
    tr, cr = best(ref_eval)
    to, co = best(lambda: orc.chi2_of(s, pv))
    print(f'cube {nx}x{nx}x{nv}, nlayer 4, K={chan.gaussbeam.shape[0]}, one core')
    print(f'reference: {tr:.3f} s per evaluation ({1 / tr:.3f} evals/s), setup {t_setup_ref:.1f} s, chi2 {cr!r}')
    print(f'oracle   : {to:.3f} s per evaluation ({1 / to:.3f} evals/s), setup {t_setup_orc:.1f} s, chi2 {co!r}')
    print(f'oracle / reference time: {to / tr:.2f}; chi2 relative difference {abs(co - cr) / abs(cr):.1e}')


if __name__ == '__main__':
    main()
