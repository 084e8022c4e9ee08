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
n = I.shape[1]
def frac(tx, ty):
    nx, ny = -(-n // tx), -(-n // ty)
    pad = np.zeros((I.shape[0], ny * ty, nx * tx), bool)
    pad[:, :n, :n] = I != 0
    t = pad.reshape(I.shape[0], ny, ty, nx, tx).any(axis=(2, 4))
    return t.mean()
for tx, ty in ((32, 8), (16, 16), (64, 64), (64, 8), (64, 1), (8, 8)):
    print(f'walker0 live fraction of {tx}x{ty} tiles: {frac(tx, ty):.4f}')
rows = (I != 0).any(axis=2)
print('live rows fraction', rows.mean())
