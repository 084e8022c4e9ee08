"""SYN-v1 synthetic channel-map cubes and walker sets.

The reference ships no cube (`testfits/test.cube.fits` is absent from the
checkout), so benchmarks and large-size tests use cubes of the named shapes
generated here: a Keplerian-disk truth model (from this package's own GPU
`cubemodel`) scaled to 50 sigma at peak plus seeded Gaussian noise.
"""
from __future__ import annotations

import contextlib
import io

import numpy as np

CONFIGS = {
    # name: (N, NV, beam major px, beam minor px)
    'C2': (256, 64, 8, 6),
    'C3': (512, 128, 8, 6),
    'C4': (1024, 256, 24, 18),
}
TRUTH = dict(Mstar=0.5, Rc=150., cs=0.3, h1=0.1, h2=0.3, pI=1., Rin=0., Ienv=0.,
             xoff=0., yoff=0., voff=0., incloff=0., paoff=0.)
SIGMA = 1.7e-3
RMAX = 200.0


def axes(N, NV):
    dpix = 2 * RMAX / N
    x = -(np.arange(N) - N / 2) * dpix        # descending, like R.A.
    y = (np.arange(N) - N / 2) * dpix
    v = (np.arange(NV) - NV / 2 + 0.5) * 0.125
    return x, y, v, dpix


def make_fit(config='C3', bpa=0.0, seed=12345, N=None, NV=None, beam_px=None,
             verbose=False, max_scratch_bytes=0, scaling='uniform', envelope=False,
             precision='f64', beam_tol=0.0):
    """Return a `ChannelFit` with a SYN-v1 cube loaded and `makegrid` done.
    The data are the same for every `scaling` and `envelope` (truth from the
    'uniform' disk-only model); `envelope=True` only switches the class flag
    (:223), which makes every sample beyond Rc emit (the dense-sky workload)."""
    from .channelfit import ChannelFit
    n, nv, bmaj_px, bmin_px = CONFIGS[config] if config in CONFIGS else (N, NV, *beam_px)
    if N is not None:
        n = N
    if NV is not None:
        nv = NV
    if beam_px is not None:
        bmaj_px, bmin_px = beam_px
    x, y, v, dpix = axes(n, nv)
    sink = contextlib.nullcontext() if verbose else contextlib.redirect_stdout(io.StringIO())
    kw = dict(pa=2.0, incl=60.0, rmax=RMAX, signmajor=1, signminor=1,
              max_scratch_bytes=max_scratch_bytes)
    with sink:
        # pass 1: data = 1 everywhere only to obtain a positively scaled truth
        fit = ChannelFit(progressbar=False)
        fit.set_cube(x, y, v, np.ones((nv, n, n)), bmaj_px * dpix, bmin_px * dpix,
                     bpa, SIGMA)
        fit.makegrid(cubefits=None, **kw)
        truth = fit.cubemodel(**TRUTH)
        truth *= 50 * SIGMA / truth.max()
        rng = np.random.default_rng(seed)
        data = truth + rng.normal(0.0, SIGMA, truth.shape)
        fit.close()
        if scaling != 'uniform' or envelope or precision != 'f64' or beam_tol:
            fit = ChannelFit(scaling=scaling, envelope=envelope, progressbar=False,
                             precision=precision, beam_tol=beam_tol)
        fit.set_cube(x, y, v, data, bmaj_px * dpix, bmin_px * dpix, bpa, SIGMA)
        fit.makegrid(cubefits=None, **kw)
    return fit


WORKLOADS = ('sparse', 'dense_disk', 'envelope')


def walkers(nw, seed=2024, workload='sparse'):
    """Natural-unit parameter rows [nw, 13].

    'sparse' (the headline): (log Mstar, log Rc, h1, h2, pI) uniform in the
    example's priors (reference example_channelfit.py:22-32), cs = 0.4,
    everything else 0.
    'dense_disk': the same with a large, broad-lined disk -- Rc in [250, 300] au,
    Mstar in [0.5, 1], cs = 1.0 km/s -- which lights most of the sky in most
    channels.
    'envelope': the 'sparse' rows with Ienv = 1, for a fit made with
    `envelope=True` (every sample beyond Rc emits; nothing is culled by radius)."""
    u = np.random.default_rng(seed).random((nw, 5))
    P = np.zeros((nw, 13))
    P[:, 0] = 10**(np.log10(0.1) + u[:, 0] * (np.log10(1.0) - np.log10(0.1)))
    P[:, 1] = 10**(np.log10(30.) + u[:, 1] * (np.log10(300.) - np.log10(30.)))
    P[:, 2] = 0.4
    P[:, 3] = 0.5 * u[:, 2]
    P[:, 4] = 0.5 * u[:, 3]
    P[:, 5] = 3.0 * u[:, 4]
    if workload == 'dense_disk':
        P[:, 0] = 0.5 + 0.5 * u[:, 0]
        P[:, 1] = 250. + 50. * u[:, 1]
        P[:, 2] = 1.0
    elif workload == 'envelope':
        P[:, 7] = 1.0
    elif workload != 'sparse':
        raise ValueError(f'workload {workload!r}: expected one of {WORKLOADS}')
    return P


def bytes_per_eval(fit, itemsize=8):
    """Algorithmic HBM bytes of one likelihood evaluation:
    write Iunif; read Iunif + write Iconv; read Iconv + read data."""
    e = fit.engine
    return itemsize * e.nv * (4 * e.nneed**2 + e.nx * e.ny)


# ---------------------------------------------------------------------------
# P5: pvfitting's mock-PV likelihood on the reference's test PV diagrams.  The
# diagrams and axes travel as data in tests/golden/pvfit_example_fit.npz (the
# FITS files themselves stay in the reference).
# ---------------------------------------------------------------------------
PV_EXAMPLE = dict(incl=85., linewidth=None, rin=1., vmask=(-0.3, 0.3), zmax=1000.,
                  nnest=[2] * 6, reslim=10, signmajor=-1, signminor=1,
                  pa_major=2.0 - 180, pa_minor=92.0 - 180)


def _pv_example_data():
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f = np.load(os.path.join(here, 'tests', 'golden', 'pvfit_example_fit.npz'))
    x, v = f['obs_x'], f['obs_v']
    dx = x[1] - x[0]
    nz = int(PV_EXAMPLE['zmax'] / dx + 0.5)
    z = (np.arange(2 * nz + 1) - nz) * dx
    beam = f['obs_beam']
    vv = np.meshgrid(x, v)[1]
    hide = (PV_EXAMPLE['vmask'][0] < vv) * (vv < PV_EXAMPLE['vmask'][1])
    major = np.where(hide, np.nan, f['obs_major'].astype(float))
    minor = np.where(hide, np.nan, f['obs_minor'].astype(float))
    rarea = np.pi / (4. * np.log(2.)) * beam[0] * beam[1] / dx / dx
    return dict(x=x, z=z, v=v, beam=beam, major=major, minor=minor,
                raw_major=f['obs_major'], raw_minor=f['obs_minor'],
                sigma=float(f['obs_sigma']), rarea=rarea)


def make_pvfit():
    """(MockPVD with observations attached, per-call keywords) for P5."""
    from .mockpvd import MockPVD
    d = _pv_example_data()
    e = PV_EXAMPLE
    m = MockPVD(d['x'], d['z'], d['v'], nnest=e['nnest'], beam=d['beam'],
                reslim=e['reslim'], signmajor=e['signmajor'], signminor=e['signminor'],
                pa_major=e['pa_major'], pa_minor=e['pa_minor'])
    m.set_observations(d['major'], d['minor'], d['sigma'], d['rarea'])
    return m, dict(incl=e['incl'], linewidth=e['linewidth'], rin=e['rin'],
                   rout=float(np.max(d['z'])))


def make_pvfit_oracle(orc, rc_first):
    """The same fit for oracle/pvfitting_oracle.py (passed in by the caller:
    bench.py's cpu_baseline leg or a test)."""
    d = _pv_example_data()
    e = PV_EXAMPLE
    s = orc.make_setup(d['x'], d['z'], d['v'], d['beam'], incl=e['incl'],
                       rc_first=rc_first, nnest=e['nnest'], reslim=e['reslim'],
                       signmajor=e['signmajor'], signminor=e['signminor'],
                       pa_major=e['pa_major'], pa_minor=e['pa_minor'],
                       linewidth=e['linewidth'], rin=e['rin'], rout=float(np.max(d['z'])))
    obs = orc.make_obs(d['x'], d['v'], d['raw_major'], d['raw_minor'], d['sigma'],
                       d['beam'][0], d['beam'][1], vmask=e['vmask'])
    return s, obs


def pv_walkers(nw, seed=2024):
    """Natural-unit rows (Mstar, Rc, alphainfall, taumax, frho, sig_mdl): the
    example's free parameters uniform in its (log10) priors
    (example_pvfitting.py:25-31), alphainfall=0.6 and sig_mdl=5 fixed."""
    u = np.random.default_rng(seed).random((nw, 4))
    lo = np.log10([0.1, 30., 1e-2, 1.])
    hi = np.log10([1.0, 300., 1e4, 1e3])
    p = 10**(lo + (hi - lo) * u)
    Q = np.empty((nw, 6))
    Q[:, 0], Q[:, 1], Q[:, 3], Q[:, 4] = p.T
    Q[:, 2], Q[:, 5] = 0.6, 5.0
    return Q
