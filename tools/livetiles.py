"""Fraction of conv tiles that are non-empty for the bench walkers (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
fit = synth.make_fit(cfg)
P = synth.walkers(128)
st = fit.engine.chi2_stats(P)
fr = st['live_tiles'] / st['tiles']
print(cfg, 'live tile fraction: mean %.3f min %.3f max %.3f' % (fr.mean(), fr.min(), fr.max()))
I = fit.engine.iunif(P[0])
print('walker0 nonzero voxel fraction', np.count_nonzero(I) / I.size)
nzc = [(I[k] != 0).any() for k in range(I.shape[0])]
print('channels with emission', sum(nzc), 'of', len(nzc))
