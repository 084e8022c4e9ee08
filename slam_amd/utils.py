"""Host-side helpers around the likelihood: coordinate rotation, the FITS cube
reader and the sampler driver.  Mirrors the reference `utils.py` interface
(`rot`, `ReadFits.read_cubefits`, `emcee_corner`) for the channelfit path;
astropy / emcee / corner are optional (absent in the build image).
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


class ReadFits:
    def read_cubefits(self, cubefits: str, center: str | None = None,
                      dist: float = 1, vsys: float = 0,
                      xmin: float | None = None, xmax: float | None = None,
                      ymin: float | None = None, ymax: float | None = None,
                      vmin: float | None = None, vmax: float | None = None,
                      xskip: int = 1, yskip: int = 1,
                      sigma: float | None = None) -> dict:
        """Read channel maps: axes in au / km s^-1 relative to `center`/`vsys`,
        cropped to the given limits (reference utils.py:178-277)."""
        d, h = fitsio.read_primary(cubefits)
        d = np.squeeze(d)
        h = dict(h)
        if center is None:
            cx, cy = 0, 0
        else:
            cx, cy = spherical_offsets(h['CRVAL1'], h['CRVAL2'],
                                       *parse_center(center))
        if sigma is None:
            sigma = np.mean([np.nanstd(d[:2]), np.std(d[-2:])])
            print(f'sigma = {sigma:.3e}')
        x = (np.arange(h['NAXIS1']) - h['CRPIX1'] + 1) * h['CDELT1']
        y = (np.arange(h['NAXIS2']) - h['CRPIX2'] + 1) * h['CDELT2']
        v = (np.arange(h['NAXIS3']) - h['CRPIX3'] + 1) * h['CDELT3']
        for ax, skip in ((1, xskip), (2, yskip)):          # pixel skipping
            crpix = int(h[f'CRPIX{ax}']) - 1
            start = crpix % skip
            h[f'CRPIX{ax}'] = (crpix - start) // skip + 1
            h[f'CDELT{ax}'] = h[f'CDELT{ax}'] * skip
            if ax == 1:
                x, d = x[start::skip], d[:, :, start::skip]
            else:
                y, d = y[start::skip], d[:, start::skip, :]
        v = v + h['CRVAL3']
        x = (x - cx) * 3600. * dist  # au
        y = (y - cy) * 3600. * dist  # au
        if h.get('CUNIT3') == 'Hz':
            if 'RESTFRQ' in h:
                restfreq = h['RESTFRQ']
            elif 'RESTFREQ' in h:
                restfreq = h['RESTFREQ']
            else:
                restfreq = np.mean(v)
                print('No rest frequency found. The middle frequency adopted.')
            v = (1. - v / restfreq) * _C_SI / 1.e3 - vsys  # km/s
        elif h.get('CUNIT3') == 'm/s':
            v = v * 1e-3 - vsys
        i0 = 0 if xmax is None else np.argmin(np.abs(x - xmax))
        i1 = len(x) - 1 if xmin is None else np.argmin(np.abs(x - xmin))
        j0 = 0 if ymin is None else np.argmin(np.abs(y - ymin))
        j1 = len(y) - 1 if ymax is None else np.argmin(np.abs(y - ymax))
        k0 = 0 if vmin is None else np.argmin(np.abs(v - vmin))
        k1 = len(v) - 1 if vmax is None else np.argmin(np.abs(v - vmax))
        x, y, v = x[i0:i1 + 1], y[j0:j1 + 1], v[k0:k1 + 1]
        d = d[k0:k1 + 1, j0:j1 + 1, i0:i1 + 1]
        self.offpix = (i0, j0, k0)
        dx, dy, dv = x[1] - x[0], y[1] - y[0], v[1] - v[0]
        if 'BMAJ' in h:
            bmaj = h['BMAJ'] * 3600. * dist  # au
            bmin = h['BMIN'] * 3600. * dist  # au
            bpa = h['BPA']  # deg
        else:
            bmaj, bmin, bpa = dy, -dx, 0
            print('No valid beam in the FITS file.')
        self.x, self.dx, self.nx = x, dx, len(x)
        self.y, self.dy, self.ny = y, dy, len(y)
        self.v, self.dv, self.nv = v, dv, len(v)
        self.data, self.header, self.sigma = d, h, sigma
        self.bmaj, self.bmin, self.bpa = bmaj, bmin, bpa
        self.beam = np.array([bmaj, bmin, bpa])
        self.cubefits, self.dist, self.vsys = cubefits, dist, vsys
        return {'x': x, 'y': y, 'v': v, 'data': d, 'header': h, 'sigma': sigma}

    def read_pvfits(self, pvfits: str,
                    dist: float = 1, vsys: float = 0,
                    xmin: float | None = None, xmax: float | None = None,
                    vmin: float | None = None, vmax: float | None = None,
                    xskip: int = 1,
                    sigma: float | None = None) -> dict:
        """Read a position-velocity diagram: offset axis in au (ascending),
        velocity in km s^-1 relative to `vsys`, cropped to the given limits
        (reference utils.py:279-352)."""
        d, h = fitsio.read_primary(pvfits)
        d = np.squeeze(d)
        h = dict(h)
        if sigma is None:
            sigma = np.mean([np.std(d[:2, 10:-10]), np.std(d[-2:, 10:-10]),
                             np.std(d[2:-2, :10]), np.std(d[2:-2, -10:])])
            print(f'sigma = {sigma:.3e}')
        x = (np.arange(h['NAXIS1']) - h['CRPIX1'] + 1) * h['CDELT1']
        v = (np.arange(h['NAXIS2']) - h['CRPIX2'] + 1) * h['CDELT2']
        crpix = int(h['CRPIX1']) - 1
        start = crpix % xskip
        x, d = x[start::xskip], d[:, start::xskip]
        h['CRPIX1'] = (crpix - start) // xskip + 1
        h['CDELT1'] = h['CDELT1'] * xskip
        v = v + h['CRVAL2']
        x = (x + h['CRVAL1']) * dist  # au
        if h.get('CUNIT2') == 'Hz':
            if 'RESTFRQ' in h:
                restfreq = h['RESTFRQ']
            elif 'RESTFREQ' in h:
                restfreq = h['RESTFREQ']
            else:
                restfreq = np.mean(v)
                print('No rest frequency found. The middle frequency adopted.')
            v = (1. - v / restfreq) * _C_SI / 1.e3 - vsys  # km/s
        elif h.get('CUNIT2') == 'm/s':
            v = v * 1e-3 - vsys
        i0 = 0 if xmin is None else np.argmin(np.abs(x - xmin))
        i1 = len(x) - 1 if xmax is None else np.argmin(np.abs(x - xmax))
        k0 = 0 if vmin is None else np.argmin(np.abs(v - vmin))
        k1 = len(v) - 1 if vmax is None else np.argmin(np.abs(v - vmax))
        x, v = x[i0:i1 + 1], v[k0:k1 + 1]
        d = d[k0:k1 + 1, i0:i1 + 1]
        self.offpix = (i0, k0)
        dx, dv = x[1] - x[0], v[1] - v[0]
        if 'BMAJ' in h:
            bmaj = h['BMAJ'] * 3600. * dist  # au
            bmin = h['BMIN'] * 3600. * dist  # au
            bpa = h['BPA']  # deg
        else:
            bmaj, bmin, bpa = dx, dx, 0
            print('No valid beam in the FITS file.')
        self.x, self.dx, self.nx = x, dx, len(x)
        self.v, self.dv, self.nv = v, dv, len(v)
        self.data, self.header, self.sigma = d, h, sigma
        self.bmaj, self.bmin, self.bpa = bmaj, bmin, bpa
        self.beam = np.array([bmaj, bmin, bpa])
        self.pvfits, self.dist, self.vsys = pvfits, dist, vsys
        return {'x': x, 'v': v, 'data': d, 'header': h, 'sigma': sigma}


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
    """Two-phase ensemble MCMC in a box prior (reference utils.py:18-119).

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
    chains = samples.copy()
    flat = samples.reshape((-1, ndim))
    plow, pmid, phigh = (np.percentile(flat, q, axis=0) for q in (16, 50, 84))
    perr = (phigh - plow) / 2.
    if range_corner is None:
        r_c = [rangelevel] * ndim if rangelevel is not None else np.transpose(bounds)
    else:
        r_c = range_corner
    if show_corner or figname is not None:
        _corner_plot(flat, popt, r_c, labels, figname, show_corner)
    if plot_chain:
        _chain_plot(chains, labels, figname, show_chain)
    out = [pmid, perr] if simpleoutput else [popt, plow, pmid, phigh]
    if return_sampler:
        return out, sampler
    return out


def dynesty_corner(bounds, log_prob_fn, args: list = [], labels: list = None,
                   figname: str = None, show_corner: bool = False,
                   return_evidence: bool = False, simpleoutput: bool = True,
                   wt_kwargs=None, vectorize: bool = False):
    """Dynamic nested sampling in a box prior (reference utils.py:122-174): same
    arguments and return values.  `vectorize=True`: `log_prob_fn` takes
    p[nw, ndim] (the GPU likelihoods do); dynesty proposes one point at a time,
    so the batch has one row.  Needs dynesty, which is optional: without it this
    raises ImportError instead of substituting another sampler."""
    try:
        from dynesty import DynamicNestedSampler as DNS
        from dynesty import utils as dyfunc
    except ImportError as e:
        raise ImportError('dynesty_corner needs the dynesty package '
                          '(not installed); emcee_corner has a built-in sampler') from e
    plim = np.array(bounds, dtype=float)
    ndim = len(plim[0])
    if vectorize:
        def lnlike(p):
            return float(np.asarray(log_prob_fn(np.atleast_2d(p), *args))[0])
    else:
        def lnlike(p):
            return log_prob_fn(p, *args)

    def ptform(u):
        return plim[0] + (plim[1] - plim[0]) * u
    dsampler = DNS(lnlike, ptform, ndim)
    dsampler.run_nested(print_progress=False, wt_kwargs=wt_kwargs)
    results = dsampler.results
    if (figname is not None) & (show_corner == True):        # noqa: E712 (as the reference)
        import matplotlib.pyplot as plt
        from dynesty import plotting as dyplot
        cfig, _ = dyplot.cornerplot(results, labels=labels, quantiles=[0.16, 0.5, 0.84])
        cfig.savefig(figname)
        plt.show()
    weights = results.importance_weights()
    quantiles = [dyfunc.quantile(samps, [0.16, 0.5, 0.84], weights=weights)
                 for samps in results.samples.T]
    if return_evidence:
        evidence = np.exp(results.logz[-1])
        print(f'Evidence: {evidence:.2e} +/- {evidence * results.logzerr[-1]:.2e}')
    if simpleoutput:
        pmid = np.array(quantiles).T[1]
        perr = np.array([(q[2] - q[0]) * 0.5 for q in quantiles])
        return [pmid, perr]
    popt = results.samples[np.argmax(results.logl), :]
    plow, pmid, phigh = np.array(quantiles).T
    return [popt, plow, pmid, phigh]


def _corner_plot(flat, popt, r_c, labels, figname, show):
    try:
        import corner
        import matplotlib.pyplot as plt
    except ImportError:
        print('corner is not installed: skipping the corner plot.')
        return
    corner.corner(flat, truths=popt, quantiles=[0.16, 0.5, 0.84],
                  show_titles=True, range=r_c, labels=labels)
    if figname is not None:
        plt.savefig(figname)
    if show:
        plt.show()
    plt.close()


def _chain_plot(chains, labels, figname, show):
    import matplotlib.pyplot as plt
    nwalkers, nsteps, ndim = chains.shape
    fig, axes = plt.subplots(ndim, 1, sharex=True, squeeze=False)
    for i, ax in enumerate(axes[:, 0]):
        ax.plot(np.arange(nsteps), chains[:, :, i].T, 'k', lw=0.5)
        if labels is not None:
            ax.set_ylabel(labels[i])
    axes[-1, 0].set_xlabel('Step number')
    if figname is not None:
        fname = (figname.replace('.corner', '.chains') if '.corner' in figname
                 else figname.replace('.png', '.chains.png'))
        fig.savefig(fname)
    if show:
        plt.show()
    plt.close()
