"""Free / fixed / log-sampled bookkeeping of a fit: which parameters the sampler moves,
how its vectors map to natural-unit parameter rows, the prior box, and the result table.
One implementation for `ChannelFit.fitting` (reference `_channelfit.py:539-663`) and
`PVFitting.fit_mockpvd` (`_pvfitting.py:153-257`), which keep these rules in-line: a fixed
parameter is given in natural units; some parameters (`logidx`) are sampled as log10; the
`.popt.txt` table has one row per parameter and the columns popt, plow, pmid, phigh.
"""
from __future__ import annotations

import numpy as np


class FitSpace:

    def __init__(self, keys, logidx, fixed_params: dict, ranges):
        self.keys = list(keys)
        self.logidx = np.asarray(logidx, dtype=int)
        given = [fixed_params.get(k) for k in self.keys]
        self.free_mask = np.array([g is None for g in given])
        self.free = {k: g is None for k, g in zip(self.keys, given)}
        self.any_free = bool(self.free_mask.any())
        # fixed values in the units the sampler's vectors use (log10 where sampled so);
        # with nothing to sample they stay natural: the all-fixed branch evaluates them as given
        self.base = np.array([np.nan if g is None else float(g) for g in given])
        box = np.array(ranges, dtype=float)
        with np.errstate(all='ignore'):
            if self.any_free:
                held = self.logidx[~self.free_mask[self.logidx]]
                self.base[held] = np.log10(self.base[held])
            box[self.logidx] = np.log10(box[self.logidx])
        self.plim = box[self.free_mask].T                    # [2, ndim]: lower, upper

    @property
    def ndim(self) -> int:
        return int(self.free_mask.sum())

    def labels(self, names):
        """Axis labels of the sampled parameters ('log' in front of the log-sampled ones)."""
        out = np.array(names, dtype=object)              # no fixed-width truncation of 'log...'
        out[self.logidx] = ['log' + out[i] for i in self.logidx]
        return out[self.free_mask]

    def to_natural(self, P):
        """Sampled vectors [nw, ndim] (or one vector) -> natural-unit rows [nw, len(keys)]."""
        P = np.atleast_2d(np.asarray(P, dtype=float))
        Q = np.tile(self.base, (len(P), 1))
        Q[:, self.free_mask] = P
        Q[:, self.logidx] = 10**Q[:, self.logidx]
        return Q

    def corner_ranges(self, spec):
        """`range_corner` given per parameter name -> the list emcee_corner takes: 0.8 where
        nothing is given; a [low, high] pair of a log-sampled parameter goes to log10, a
        float (a quantile level) stays as it is."""
        if type(spec) is not dict:
            return spec
        out = []
        for i, k in enumerate(self.keys):
            r = spec[k] if k in spec else 0.8
            if i in self.logidx and type(r) is not float:
                r = np.log10(r)
            if self.free[k]:
                out.append(r)
        return out

    def report(self, plow, popt, phigh, digits, units, formats=None):
        """One line per parameter: name = low, best, high unit."""
        formats = formats or ['f'] * len(self.keys)
        for i, (k, d, u, f) in enumerate(zip(self.keys, digits, units, formats)):
            print(f'{k} = {plow[i]:.{d:d}{f}}, {popt[i]:.{d:d}{f}}, {phigh[i]:.{d:d}{f}} {u}')

    def write_table(self, path, popt, plow, pmid, phigh):
        """<filename>.popt.txt as the reference writes it."""
        with open(path, 'w') as f:
            f.write('#Rows:' + ','.join(self.keys) + '\n')
            f.write('#Columns:' + ','.join(['popt', 'plow', 'pmid', 'phigh']) + '\n')
            np.savetxt(f, np.transpose([popt, plow, pmid, phigh]))
