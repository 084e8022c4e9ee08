#!/usr/bin/env python3
"""Generate tests/golden/pvfit_*.npz by RUNNING THE REFERENCE's pvfitting path.

Build-container only: imports `/root/reference` (jinshisai/SLAM) with the inert
third-party stubs of `_refstubs.py` (numba's jit -> identity, astropy constants
-> the CODATA/IAU numbers, `astropy.io.fits.open` -> our own minimal FITS
reader), drives `PVFitting.put_PV` / `MockPVD.generate_mockpvd` /
`PVFitting.fit_mockpvd` on the reference's test.pvfitting.{major,minor}.fits
and on small seeded synthetic grids, and stores inputs + outputs as .npz.
No reference source is copied; the vectors are data.  The GPU box never runs
this script -- it only reads the .npz files.

Usage:  python tests/golden/make_golden_pv.py
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _refstubs  # noqa: E402

_refstubs.install()
from slam_amd import fitsio  # noqa: E402


class _HDU:
    def __init__(self, path):
        self.data, self.header = fitsio.read_primary(path)


sys.modules['astropy.io.fits'].open = lambda p: [_HDU(p)]
sys.path.insert(0, '/root/reference')
import matplotlib  # noqa: E402

matplotlib.use('Agg')
import pvfitting._pvfitting as pvf  # noqa: E402
import pvfitting.mockpvd as mp  # noqa: E402
from pvfitting import precalculation as pc  # noqa: E402
from pvfitting import PVFitting  # noqa: E402
from pvfitting.mockpvd import MockPVD  # noqa: E402

AXES = ('major', 'minor')


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def reset_reference_caches():
    """The reference keeps per-grid caches in module globals and never
    invalidates them; a new configuration needs a clean slate."""
    for d in (pc.elos_r, pc.elos_t, pc.elos_p, pc.t, pc.idx_t, pc.r_org, pc.j_org):
        for a in AXES:
            d[a] = [None] * pc.lmax
    pc.gauss_xy = pc.gauss_v = pc.vedge = None


class Taps:
    """Record rho2tau outputs and the arrays handed to scipy's convolve."""

    def __enter__(self):
        self.tau, self.conv_in = [], []
        self._r2t, self._conv = mp.rho2tau, mp.convolve

        def r2t(vlos, rho):
            out = self._r2t(vlos, rho)
            self.tau.append(out.copy())
            return out

        def conv(a, g, mode='same'):
            self.conv_in.append(np.array(a))
            return self._conv(a, g, mode=mode)
        mp.rho2tau, mp.convolve = r2t, conv
        return self

    def __exit__(self, *exc):
        mp.rho2tau, mp.convolve = self._r2t, self._conv


def record_tables(outdir):
    rec = dict(Nr=pc.Nr, Ntheta=pc.Ntheta, lnr0=pc.lnr0, dlnr=pc.dlnr,
               theta0=pc.theta0, dtheta=pc.dtheta)
    for k in ('rho_env', 'rho_disk', 'vr_env', 'vt_env', 'vp_all'):
        a = getattr(pc, k)
        rec[k + '_sum'] = a.sum()
        rec[k + '_abssum_rows'] = np.abs(a).sum(axis=1)
        rec[k + '_stride'] = a[::7, ::13].copy()
    np.savez_compressed(os.path.join(outdir, 'pvfit_tables.npz'), **rec)


def record_model(name, outdir, x, z, v, beam, incl, vectors, full_cache,
                 tap_vectors, obs=None, lnp_sig=5.0, vmask=(0., 0.), **kw):
    """One MockPVD configuration: inputs, grid, caches (full or sampled),
    per-vector PVs and taps; optional likelihood values against `obs`."""
    reset_reference_caches()
    m = MockPVD(x, z, v, beam=beam, **kw)
    g = m.grid
    rec = dict(in_x=x, in_z=z, in_v=v, in_incl=incl,
               in_beam=np.array(beam if beam is not None else [np.nan] * 3),
               in_rc_first=vectors[0][1],
               in_kw=np.array(repr(sorted(kw.items()))))
    for k, val in kw.items():
        rec['in_' + k] = np.array(np.nan if val is None else val)
    rout = np.max(z)
    rec['in_rout'] = rout
    rec['in_linewidth'] = np.array(np.nan if LINEWIDTH[name] is None else LINEWIDTH[name])
    rec['grid_nlevels'] = g.nlevels
    rec['grid_shapes'] = np.array(g.ngrids)
    rec['grid_y'] = m.y if beam is not None else np.array([-(m.x[1] - m.x[0]), 0., m.x[1] - m.x[0]])
    rec['grid_box'] = np.array([[g.xinest[l], g.yinest[l], g.zinest[l]]
                                for l in range(1, g.nlevels)], dtype=np.int64).reshape(-1, 3, 2)
    rec['grid_ncell'] = np.array([a.size for a in g.xnest])
    for l in range(g.nlevels):
        for c, a in zip('xyz', (g.xaxes[l], g.yaxes[l], g.zaxes[l])):
            rec[f'grid_axis_{c}{l}'] = a
    rec['set_signs'] = np.array([m.signmajor, m.signminor])
    rec['set_iradshift'] = m.iradshift
    names = []
    for vi, q in enumerate(vectors):
        Mstar, Rc, alpha, taumax, frho = q
        with Taps() as t:
            out = m.generate_mockpvd(Mstar, Rc, alphainfall=alpha, taumax=taumax,
                                     frho=frho, incl=incl,
                                     linewidth=LINEWIDTH[name], rout=rout,
                                     axis='both')
        nm = f'v{vi}'
        names.append(nm)
        rec[f'vec_{nm}_q'] = np.array(q, dtype=float)
        rec[f'vec_{nm}_major'], rec[f'vec_{nm}_minor'] = out
        if vi in tap_vectors:
            L = g.nlevels
            for ai, a in enumerate(AXES):
                # call order inside generate_pvd: innermost level first
                for k in range(L):
                    rec[f'vec_{nm}_own_{a}_{L - 1 - k}'] = t.tau[ai * L + k]
            per_axis = len(t.conv_in) // 2
            for ai, a in enumerate(AXES):
                for k in range(per_axis):
                    rec[f'vec_{nm}_convin_{a}_{k}'] = t.conv_in[ai * per_axis + k]
    rec['vector_names'] = np.array(names)
    # caches as the first evaluation left them
    for a in AXES:
        for l in range(g.nlevels):
            for k, d in (('idx_t', pc.idx_t), ('j_org', pc.j_org), ('r_org', pc.r_org),
                         ('elos_r', pc.elos_r), ('elos_t', pc.elos_t),
                         ('elos_p', pc.elos_p)):
                arr = np.asarray(d[a][l])
                if full_cache:
                    rec[f'cache_{a}_{l}_{k}'] = arr
                else:
                    rec[f'cache_{a}_{l}_{k}_sum'] = arr.sum()
                    rec[f'cache_{a}_{l}_{k}_stride'] = arr[::97].copy()
    if pc.gauss_xy is not None:
        rec['set_gauss_xy'] = pc.gauss_xy
    if pc.gauss_v is not None:
        rec['set_gauss_v'] = pc.gauss_v
    rec['set_vedge'] = pc.vedge
    np.savez_compressed(os.path.join(outdir, f'pvfit_{name}.npz'), **rec)
    print(name, 'levels', g.nlevels, 'cells', rec['grid_ncell'].sum(),
          'max major', [float(np.nanmax(rec[f'vec_{n}_major'])) for n in names][:3])
    return m


LINEWIDTH = {}


def capture_fit(pv, fit_kw, npoint=5, seed=11):
    """Run the reference's fit_mockpvd with a stand-in sampler that records the
    prior box, the labels and lnprob on seeded points."""
    cap = {}

    def fake_emcee_corner(plim, lnprob, simpleoutput=True, **kw):
        plim = np.array(plim)
        rng = np.random.default_rng(seed)
        P = plim[0] + (plim[1] - plim[0]) * rng.random((npoint, plim.shape[1]))
        cap['plim'], cap['P'] = plim, P
        cap['labels'] = np.array(kw['labels'])
        cap['lnp'] = np.array([lnprob(p) for p in P])
        return [P[0], P[1], P[2], P[3]]

    real = pvf.emcee_corner
    pvf.emcee_corner = fake_emcee_corner
    cwd = os.getcwd()
    try:
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            quiet(pv.fit_mockpvd, filename='gold', progressbar=False, **fit_kw)
            with open('gold.popt.txt') as f:
                cap['popt_txt'] = np.array(f.read())
    finally:
        os.chdir(cwd)
        pvf.emcee_corner = real
    cap['popt'] = np.array([pv.popt[k] for k in pv.popt], dtype=float)
    cap['plow'] = np.array([pv.plow[k] for k in pv.plow], dtype=float)
    return cap


def record_example(outdir):
    """The reference's own example (example_pvfitting.py) on its test FITS."""
    tf = '/root/reference/testfits/test.pvfitting.%s.fits'
    pv = PVFitting()
    quiet(pv.put_PV, pvmajorfits=tf % 'major', pvminorfits=tf % 'minor',
          dist=140, vsys=5.9, rmax=200, vmin=-3.6, vmax=3.6, sigma=1.7e-3,
          skipto=5)
    x, v = pv.x, pv.v
    dx = x[1] - x[0]
    nz = int(1000 / dx + 0.5)
    z = (np.arange(2 * nz + 1) - nz) * dx
    vectors = [(0.5, 150., 0.6, 10., 10.), (0.2, 40., 0.6, 1e-2, 1.),
               (0.9, 290., 1.0, 1e4, 1e3), (0.35, 90., 0.0, 3., 50.),
               (0.7, 210., 0.3, 100., 300.)]
    LINEWIDTH['example'] = None
    record_model('example', outdir, x, z, v, pv.beam, 85., vectors,
                 full_cache=False, tap_vectors=(0,), nnest=[2] * 6, nsubgrid=1,
                 reslim=10, signmajor=-1, signminor=1, pa_major=2.0 - 180,
                 pa_minor=92.0 - 180)
    f = np.load(os.path.join(outdir, 'pvfit_example.npz'))
    assert float(np.nanmax(f['vec_v0_major'])) == 0.4066004035049253
    assert float(np.nansum(f['vec_v0_major'])) == 163.9387398678693
    assert float(np.nansum(f['vec_v0_minor'])) == 59.05831187700667
    print('known answers reproduced')
    # likelihood closure, as the example calls it (first Rc = whatever the
    # stand-in sampler draws first -> recorded as fit_rc_first)
    reset_reference_caches()
    fit_kw = dict(Mstar_range=[0.1, 1.0], Rc_range=[30, 300],
                  taumax_range=[1e-2, 1e4], frho_range=[1, 1e3],
                  fixed_params={'alphainfall': 0.6, 'sig_mdl': 5},
                  show=False, vmask=[-0.3, 0.3], incl=85, pa_major=2.0 - 180,
                  pa_minor=92.0 - 180, signmajor=-1, signminor=1,
                  kwargs_emcee_corner={'nwalkers_per_ndim': 4, 'nburnin': 3,
                                       'nsteps': 3, 'rangelevel': 0.99},
                  n_nest=[2] * 6, reslim=10, zmax=1000)
    cap = capture_fit(pv, fit_kw)
    rec = {'fit_' + k: val for k, val in cap.items()}
    rec['fit_rc_first'] = 10 ** cap['P'][0][1]
    # the all-fixed branch (reduced chi2, no log term); caches stay as they are
    pv2 = PVFitting()
    for k in ('x', 'v', 'dx', 'dv', 'sigma', 'bmaj', 'bmin', 'bpa', 'beam',
              'dpvmajor', 'dpvminor', 'header', 'offpix', 'vorg'):
        setattr(pv2, k, getattr(pv, k))
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            quiet(pv2.fit_mockpvd, filename='g2', progressbar=False,
                  fixed_params={'Mstar': 0.5, 'Rc': 150., 'alphainfall': 0.6,
                                'taumax': 10., 'frho': 10., 'sig_mdl': 5.},
                  **{k: fit_kw[k] for k in ('vmask', 'incl', 'pa_major', 'pa_minor',
                                            'signmajor', 'signminor', 'n_nest',
                                            'reslim', 'zmax', 'show')})
        finally:
            os.chdir(cwd)
    rec['fit_chi2r_allfixed'] = np.float64(pv2.chi2r)
    rec.update(obs_x=x, obs_v=v, obs_major=pv.dpvmajor, obs_minor=pv.dpvminor,
               obs_sigma=pv.sigma, obs_beam=pv.beam, obs_dx=pv.dx, obs_dv=pv.dv,
               obs_vorg=pv.vorg, obs_offpix=np.array(pv.offpix),
               obs_getquad=np.array([pvf.getquad(pv.dpvmajor),
                                     pvf.getquad(pv.dpvminor)]))
    np.savez_compressed(os.path.join(outdir, 'pvfit_example_fit.npz'), **rec)
    print('example fit capture: lnp =', cap['lnp'], 'chi2r', pv2.chi2r)


def record_small(outdir):
    rng = np.random.default_rng(5)
    # S1: odd axes (a cell sits exactly on the star), spectral smoothing,
    # rotated beam, flipped minor sign
    x = (np.arange(15) - 7) * 10.
    z = (np.arange(21) - 10) * 10.
    v = (np.arange(12) - 5.5) * 0.4
    vec = [(0.4, 60., 0.7, 5., 20.), (0.8, 25., 1.0, 0.5, 1.),
           (0.15, 140., 0.2, 200., 500.)]
    LINEWIDTH['S1'] = 0.5
    record_model('S1', outdir, x, z, v, [25., 18., 30.], 60., vec,
                 full_cache=True, tap_vectors=(0, 1, 2), nnest=[2, 2],
                 nsubgrid=1, reslim=3, signmajor=1, signminor=-1, pa_major=10.,
                 pa_minor=100.)
    # S2: even axes, sub-gridded x/z, refinement factor 3 then 2
    x = (np.arange(12) - 5.5) * 8.
    z = (np.arange(16) - 7.5) * 8.
    v = (np.arange(9) - 4) * 0.5
    vec = [(0.5, 50., 0.6, 10., 10.), (0.3, 110., 0.9, 1., 100.)]
    LINEWIDTH['S2'] = None
    record_model('S2', outdir, x, z, v, [20., 14., -15.], 75., vec,
                 full_cache=True, tap_vectors=(0, 1), nnest=[3, 2], nsubgrid=2,
                 reslim=2, signmajor=-1, signminor=-1, pa_major=0., pa_minor=90.)
    # S3: no nesting, no beam
    x = (np.arange(11) - 5) * 12.
    z = (np.arange(13) - 6) * 12.
    v = (np.arange(10) - 4.5) * 0.3
    vec = [(0.6, 70., 0.5, 2., 5.)]
    LINEWIDTH['S3'] = None
    record_model('S3', outdir, x, z, v, None, 45., vec, full_cache=True,
                 tap_vectors=(0,), nnest=None, nsubgrid=1, reslim=10,
                 signmajor=1, signminor=1, pa_major=0., pa_minor=90.)
    del rng


def main():
    record_tables(HERE)
    record_example(HERE)
    record_small(HERE)


if __name__ == '__main__':
    main()
