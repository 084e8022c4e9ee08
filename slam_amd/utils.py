"""Host-side helpers around the likelihood: coordinate rotation, the FITS cube
reader and the sampler driver.  Mirrors the reference `utils.py` interface
(`rot`, `ReadFits.read_cubefits`, `emcee_corner`, `dynesty_corner`) for the
channelfit path; astropy / emcee / corner / dynesty are optional (absent in the
build image).
"""
from __future__ import annotations

import re

import numpy as np

from . import fitsio

_C_SI = 299792458.0


def gauss1d(x, amp, mean, fwhm):
    """utils.py:14-15"""
    return amp * np.exp2(-4. * ((x - mean) / fwhm)**2)


def rot(x, y, pa):
    """Rotate to (minor-axis, major-axis) coordinates (utils.py:354-357)."""
    c, s = np.cos(pa), np.sin(pa)
    return np.array([x * c - y * s, x * s + y * c])


# ---------------------------------------------------------------------------
# FITS cube reader
# ---------------------------------------------------------------------------
_HMS = re.compile(r'^\s*([+-]?\d+)h(\d+)m([\d.]+)s\s+([+-]?\d+)d(\d+)m([\d.]+)s\s*$')


def parse_center(center: str):
    """'04h39m53.878s +26d03m09.43s' -> (ra_deg, dec_deg)."""
    m = _HMS.match(center)
    if not m:
        raise ValueError(f'cannot parse coordinates {center!r}')
    h, mi, s, d, dm, ds = m.groups()
    ra = (abs(int(h)) + int(mi) / 60 + float(s) / 3600) * 15
    sign = -1 if d.strip().startswith('-') else 1
    dec = sign * (abs(int(d)) + int(dm) / 60 + float(ds) / 3600)
    return ra, dec


def spherical_offsets(ra0, dec0, ra, dec):
    """Longitude/latitude [deg] of (ra, dec) in the sky-offset frame whose
    origin is (ra0, dec0) -- what `SkyCoord.spherical_offsets_to` returns
    (utils.py:211-216)."""
    a0, d0, a, d = np.radians([ra0, dec0, ra, dec])
    da = a - a0
    x = np.cos(d) * np.cos(da) * np.cos(d0) + np.sin(d) * np.sin(d0)
    y = np.cos(d) * np.sin(da)
    z = -np.cos(d) * np.cos(da) * np.sin(d0) + np.sin(d) * np.cos(d0)
    return np.degrees(np.arctan2(y, x)), np.degrees(np.arcsin(z))


def _pixel_axis(h, ax):
    """World offsets of FITS axis `ax` from its reference pixel (CDELT units)."""
    return (np.arange(h[f'NAXIS{ax}']) - h[f'CRPIX{ax}'] + 1) * h[f'CDELT{ax}']


def _decimate(h, ax, coord, data, data_axis, skip):
    """Keep every `skip`-th pixel of FITS axis `ax`, phased so that the
    reference pixel survives; the header's CRPIX / CDELT follow."""
    ref = int(h[f'CRPIX{ax}']) - 1
    first = ref % skip
    h[f'CRPIX{ax}'] = (ref - first) // skip + 1
    h[f'CDELT{ax}'] = h[f'CDELT{ax}'] * skip
    pick = [slice(None)] * data.ndim
    pick[data_axis] = slice(first, None, skip)
    return coord[first::skip], data[tuple(pick)]


def _to_velocity(h, ax, w, vsys):
    """Spectral axis -> km/s relative to vsys (radio convention for 'Hz')."""
    unit = h.get(f'CUNIT{ax}')
    if unit == 'Hz':
        rest = h.get('RESTFRQ', h.get('RESTFREQ'))
        if rest is None:
            rest = np.mean(w)
            print('No rest frequency found. The middle frequency adopted.')
        return (1. - w / rest) * _C_SI / 1.e3 - vsys
    if unit == 'm/s':
        return w * 1e-3 - vsys
    return w


def _nearest(coord, value, default):
    return default if value is None else int(np.argmin(np.abs(coord - value)))


def _beam(h, dist, fallback):
    """(bmaj, bmin, bpa) in au / deg from the header, else `fallback`."""
    if 'BMAJ' in h:
        return h['BMAJ'] * 3600. * dist, h['BMIN'] * 3600. * dist, h['BPA']
    print('No valid beam in the FITS file.')
    return fallback


class ReadFits:
    """FITS input of the analysis classes: same methods, arguments, attributes
    and return dicts as the reference's `ReadFits` (utils.py:177-352), read
    with this package's own primary-HDU reader (astropy is optional)."""

    def read_cubefits(self, cubefits: str, center: str | None = None,
                      dist: float = 1, vsys: float = 0,
                      xmin: float | None = None, xmax: float | None = None,
                      ymin: float | None = None, ymax: float | None = None,
                      vmin: float | None = None, vmax: float | None = None,
                      xskip: int = 1, yskip: int = 1,
                      sigma: float | None = None) -> dict:
        """Channel maps with axes in au / km s^-1 relative to `center` / `vsys`,
        decimated by xskip / yskip and cropped to the limits; x follows the
        file's (descending R.A.) order, so `xmax` is the first column kept."""
        cube, h = fitsio.read_primary(cubefits)
        cube, h = np.squeeze(cube), dict(h)
        off = (0, 0) if center is None else spherical_offsets(
            h['CRVAL1'], h['CRVAL2'], *parse_center(center))
        if sigma is None:                  # rms of the two edge channels at either end
            sigma = np.mean([np.nanstd(cube[:2]), np.std(cube[-2:])])
            print(f'sigma = {sigma:.3e}')
        x, y, v = (_pixel_axis(h, ax) for ax in (1, 2, 3))
        x, cube = _decimate(h, 1, x, cube, 2, xskip)
        y, cube = _decimate(h, 2, y, cube, 1, yskip)
        x = (x - off[0]) * 3600. * dist
        y = (y - off[1]) * 3600. * dist
        v = _to_velocity(h, 3, v + h['CRVAL3'], vsys)
        lo = (_nearest(x, xmax, 0), _nearest(y, ymin, 0), _nearest(v, vmin, 0))
        hi = (_nearest(x, xmin, len(x) - 1), _nearest(y, ymax, len(y) - 1),
              _nearest(v, vmax, len(v) - 1))
        x, y, v = (c[a:b + 1] for c, a, b in zip((x, y, v), lo, hi))
        cube = cube[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1]
        self.offpix = lo
        self.x, self.y, self.v, self.data = x, y, v, cube
        self.dx, self.dy, self.dv = x[1] - x[0], y[1] - y[0], v[1] - v[0]
        self.nx, self.ny, self.nv = len(x), len(y), len(v)
        self.header, self.sigma = h, sigma
        self.bmaj, self.bmin, self.bpa = _beam(h, dist, (self.dy, -self.dx, 0))
        self.beam = np.array([self.bmaj, self.bmin, self.bpa])
        self.cubefits, self.dist, self.vsys = cubefits, dist, vsys
        return {'x': x, 'y': y, 'v': v, 'data': cube, 'header': h, 'sigma': sigma}

    def read_pvfits(self, pvfits: str,
                    dist: float = 1, vsys: float = 0,
                    xmin: float | None = None, xmax: float | None = None,
                    vmin: float | None = None, vmax: float | None = None,
                    xskip: int = 1,
                    sigma: float | None = None) -> dict:
        """Position-velocity diagram: offset axis in au (ascending), velocity in
        km s^-1 relative to `vsys`, decimated by xskip and cropped."""
        pv, h = fitsio.read_primary(pvfits)
        pv, h = np.squeeze(pv), dict(h)
        if sigma is None:                  # rms of the four borders of the diagram
            sigma = np.mean([np.std(pv[:2, 10:-10]), np.std(pv[-2:, 10:-10]),
                             np.std(pv[2:-2, :10]), np.std(pv[2:-2, -10:])])
            print(f'sigma = {sigma:.3e}')
        x, v = _pixel_axis(h, 1), _pixel_axis(h, 2)
        x, pv = _decimate(h, 1, x, pv, 1, xskip)
        x = (x + h['CRVAL1']) * dist
        v = _to_velocity(h, 2, v + h['CRVAL2'], vsys)
        lo = (_nearest(x, xmin, 0), _nearest(v, vmin, 0))
        hi = (_nearest(x, xmax, len(x) - 1), _nearest(v, vmax, len(v) - 1))
        x, v = x[lo[0]:hi[0] + 1], v[lo[1]:hi[1] + 1]
        pv = pv[lo[1]:hi[1] + 1, lo[0]:hi[0] + 1]
        self.offpix = lo
        self.x, self.v, self.data = x, v, pv
        self.dx, self.dv = x[1] - x[0], v[1] - v[0]
        self.nx, self.nv = len(x), len(v)
        self.header, self.sigma = h, sigma
        self.bmaj, self.bmin, self.bpa = _beam(h, dist, (self.dx, self.dx, 0))
        self.beam = np.array([self.bmaj, self.bmin, self.bpa])
        self.pvfits, self.dist, self.vsys = pvfits, dist, vsys
        return {'x': x, 'v': v, 'data': pv, 'header': h, 'sigma': sigma}


# ---------------------------------------------------------------------------
# Sampler
# ---------------------------------------------------------------------------
class StretchSampler:
    """Goodman & Weare (2010) affine-invariant stretch move with a vectorised
    log-probability: the ensemble is split in two halves and each half is
    proposed and evaluated as ONE batch (the update emcee's default
    `StretchMove` performs).  Exposes `.chain` [nwalkers, nsteps, ndim] and
    `.lnprobability` [nwalkers, nsteps] like emcee 2 (utils.py:61, :71)."""

    def __init__(self, nwalkers, ndim, log_prob_vec, a=2.0, rng=None):
        if nwalkers < 2 * ndim or nwalkers % 2:
            raise ValueError('need an even number of walkers >= 2*ndim')
        self.nwalkers, self.ndim, self.a = nwalkers, ndim, a
        self.log_prob_vec = log_prob_vec
        self.rng = np.random.default_rng() if rng is None else rng
        self.naccepted = np.zeros(nwalkers)
        self.ncalls = 0

    def _lnp(self, P):
        self.ncalls += len(P)
        lp = np.asarray(self.log_prob_vec(P), dtype=float)
        if lp.shape != (len(P),):
            raise ValueError('log-probability must return one value per row')
        if np.any(np.isnan(lp)):
            raise ValueError('log-probability returned NaN')
        return lp

    def run_mcmc(self, p0, nsteps):
        p = np.array(p0, dtype=float)
        lnp = self._lnp(p)
        nw, nd = self.nwalkers, self.ndim
        self.chain = np.empty((nw, nsteps, nd))
        self.lnprobability = np.empty((nw, nsteps))
        half = nw // 2
        sets = (np.arange(half), np.arange(half, nw))
        for it in range(nsteps):
            for s in (0, 1):
                S, Cc = sets[s], sets[1 - s]
                z = ((self.a - 1.) * self.rng.random(len(S)) + 1)**2 / self.a
                partner = p[Cc[self.rng.integers(len(Cc), size=len(S))]]
                q = partner - (partner - p[S]) * z[:, None]
                lnq = self._lnp(q)
                with np.errstate(all='ignore'):
                    lnr = (nd - 1.) * np.log(z) + lnq - lnp[S]
                acc = np.log(self.rng.random(len(S))) < lnr
                idx = S[acc]
                p[idx], lnp[idx] = q[acc], lnq[acc]
                self.naccepted[idx] += 1
            self.chain[:, it], self.lnprobability[:, it] = p, lnp
        return p, lnp


def emcee_corner(bounds, log_prob_fn, args: list = [],
                 nwalkers_per_ndim: int = 16,
                 nburnin: int = 2000, nsteps: int = 2000,
                 gr_check: bool = False, ndata: int = 1000,
                 labels: list = None, rangelevel: float = 0.8,
                 range_corner: list | None = None,
                 figname: str = None, show_corner: bool = False,
                 plot_chain: bool = False, show_chain: bool = False,
                 ncore: int = 1, simpleoutput: bool = True,
                 moves=None, vectorize: bool = False, seed=None,
                 return_sampler: bool = False):
    """Two-phase ensemble MCMC in a box prior (reference utils.py:18-119; same
    arguments and return values, `ncore` and the figure arguments are accepted
    and not acted on: the likelihood is batched on the GPU, figures are out of scope).

    `vectorize=True`: `log_prob_fn` takes p[nw, ndim] and returns lnp[nw]; the
    strict box prior (utils.py:32-36) is applied here and only in-prior rows
    are sent on.  With `vectorize=False` the function is called per walker,
    as in the reference.  Uses emcee when importable, else `StretchSampler`.
    """
    plim = np.array(bounds, dtype=float)
    ndim = len(plim[0])
    nwalkers = ndim * nwalkers_per_ndim
    rng = np.random.default_rng(seed)

    def lnL_vec(P):
        P = np.atleast_2d(P)
        inside = np.all((plim[0] < P) * (P < plim[1]), axis=1)
        out = np.full(len(P), -np.inf)
        if inside.any():
            if vectorize:
                out[inside] = log_prob_fn(P[inside], *args)
            else:
                out[inside] = [log_prob_fn(p, *args) for p in P[inside]]
        return out

    def gelman_rubin(samples):                         # utils.py:38-45
        n = len(samples[0])
        B = np.std(np.mean(samples, axis=1), axis=0)
        W = np.mean(np.std(samples, axis=1), axis=0)
        V = (n - 1) / n * W + (nwalkers + 1) / (nwalkers - 1) * B
        dd = ndata - ndim - 1
        return np.sqrt((dd + 3) / (dd + 1) * V / W)

    try:
        import emcee
    except ImportError:
        emcee = None

    p0 = plim[0] + (plim[1] - plim[0]) * rng.random((nwalkers, ndim))
    converge, GR = True, None
    for n in [nburnin, nsteps]:
        converge = True
        if emcee is not None:
            kw = {} if moves is None else {'moves': moves}
            sampler = emcee.EnsembleSampler(nwalkers, ndim, lnL_vec,
                                            vectorize=True, **kw)
            sampler.run_mcmc(p0, n)
            samples = sampler.get_chain().transpose(1, 0, 2)
            lnp = sampler.get_log_prob().T
        else:
            sampler = StretchSampler(nwalkers, ndim, lnL_vec, rng=rng)
            sampler.run_mcmc(p0, n)
            samples, lnp = sampler.chain, sampler.lnprobability
        if gr_check:
            GR = gelman_rubin(samples)
            if np.max(GR) > 1.25:
                converge = False
        p0 = samples[:, -1, :]
    if not converge:
        print('\nWARNING: emcee did not converge (Gelman-Rubin =',
              np.round(GR, 2), '> 1.25).\n')
    popt = samples[np.unravel_index(np.argmax(lnp), lnp.shape)]
    flat = samples.reshape((-1, ndim))
    plow, pmid, phigh = (np.percentile(flat, q, axis=0) for q in (16, 50, 84))
    perr = (phigh - plow) / 2.
    if show_corner or plot_chain:
        # corner / chain figures are not part of this package (SURVEY.md section 2, rows 5
        # and 7): the samples are returned; plot them with the reference's own helpers
        print('slam_amd.emcee_corner: figures are not drawn here (figname / show_* ignored).')
    out = [pmid, perr] if simpleoutput else [popt, plow, pmid, phigh]
    if return_sampler:
        return out, sampler
    return out


class BatchedPool:
    """`pool=` object for dynesty: its `map` evaluates a whole queue of points
    with ONE call of a vectorised log-likelihood where dynesty maps the
    likelihood itself over points (the initial live points, and every batch of
    a sampler that hands the likelihood to the pool), and falls back to an
    ordinary serial map for anything else it is asked to map (dynesty's
    point-evolution tasks, which call the scalar likelihood themselves).
    `size` is the queue length dynesty fills before it calls `map`."""

    def __init__(self, log_prob_vec, scalar_fn, size):
        self.log_prob_vec, self.scalar_fn, self.size = log_prob_vec, scalar_fn, int(size)
        self.batched_calls = self.serial_calls = 0

    def map(self, fn, items):
        items = list(items)
        inner = getattr(fn, 'func', fn)          # dynesty wraps user functions; `.func` is the original
        if inner is self.scalar_fn and len(items):
            self.batched_calls += 1
            return list(np.asarray(self.log_prob_vec(np.asarray(items, dtype=float)), dtype=float))
        self.serial_calls += 1
        return [fn(it) for it in items]


def dynesty_corner(bounds,
                   log_prob_fn, args: list = [],
                   labels: list = None,
                   figname: str = None,
                   show_corner: bool = False,
                   return_evidence: bool = False,
                   simpleoutput: bool = True,
                   wt_kwargs=None,
                   vectorize: bool = False, queue_size: int | None = None,
                   sampler_cls=None, quantile_fn=None):
    """Dynamic nested sampling in a box prior (reference utils.py:122-174; same
    arguments and return values; the corner figure is not drawn).

    `vectorize=True`: `log_prob_fn` takes p[nw, ndim] and returns lnp[nw] (the
    batched GPU likelihood).  dynesty itself asks for one point at a time, so the
    sampler gets a scalar wrapper (one-row batches) plus a `BatchedPool` of
    `queue_size` (default 8 x ndim) slots, through which the live points and any
    likelihood batch dynesty hands to its pool are evaluated as one call.

    dynesty is not importable in the build image: its results are PARITY UNPINNED
    here (tests drive this function with a stub sampler through `sampler_cls` /
    `quantile_fn`).  Without dynesty and without `sampler_cls` an ImportError says so.
    """
    ndim = len(bounds[0])
    plim = np.array(bounds, dtype=float)
    if sampler_cls is None:
        try:
            from dynesty import DynamicNestedSampler as sampler_cls
            from dynesty import utils as dyfunc
        except ImportError as e:
            raise ImportError(
                'slam_amd.utils.dynesty_corner needs the dynesty package (not installed); '
                'use emcee_corner (built-in stretch-move sampler) or pass sampler_cls') from e
        if quantile_fn is None:
            quantile_fn = dyfunc.quantile
    if quantile_fn is None:
        quantile_fn = weighted_quantile

    if vectorize:
        def lnlike(p):
            return float(np.asarray(log_prob_fn(np.asarray(p, dtype=float)[None, :], *args))[0])

        def lnlike_vec(P):
            return log_prob_fn(P, *args)
    else:
        def lnlike(p):
            return log_prob_fn(p, *args)

        def lnlike_vec(P):
            return [log_prob_fn(p, *args) for p in P]

    def ptform(u):
        return plim[0] + (plim[1] - plim[0]) * u

    kw = {}
    pool = None
    if vectorize:
        pool = BatchedPool(lnlike_vec, lnlike, queue_size or 8 * ndim)
        kw = {'pool': pool, 'queue_size': pool.size}
    dsampler = sampler_cls(lnlike, ptform, ndim, **kw)
    dsampler.run_nested(print_progress=False, wt_kwargs=wt_kwargs)
    results = dsampler.results
    if show_corner or figname is not None:
        print('slam_amd.dynesty_corner: figures are not drawn here (figname / show_corner ignored).')
    weights = results.importance_weights()
    samples = np.asarray(results.samples)
    quantiles = [quantile_fn(samps, [0.16, 0.5, 0.84], weights=weights) for samps in samples.T]
    if return_evidence:
        evidence = np.exp(results.logz[-1])
        evid_err = evidence * results.logzerr[-1]
        print(f'Evidence: {evidence:.2e} +/- {evid_err:.2e}')
    if simpleoutput:
        pmid = np.array(quantiles).T[1]
        perr = np.array([(q[2] - q[0]) * 0.5 for q in quantiles])
        return [pmid, perr]
    popt = samples[np.argmax(results.logl), :]
    plow, pmid, phigh = np.array(quantiles).T
    return [popt, plow, pmid, phigh]


def weighted_quantile(x, q, weights=None):
    """Quantiles of weighted samples: the sorted samples against the cumulative
    weights that precede them, normalised to end at 1, linearly interpolated --
    the rule of `dynesty.utils.quantile`, used when dynesty's own is absent."""
    x = np.asarray(x, dtype=float)
    q = np.atleast_1d(q)
    if weights is None:
        return np.percentile(x, list(100.0 * q)).tolist()
    order = np.argsort(x)
    cdf = np.cumsum(np.asarray(weights, dtype=float)[order])[:-1]
    cdf = np.append(0, cdf / cdf[-1])
    return np.interp(q, cdf, x[order]).tolist()
