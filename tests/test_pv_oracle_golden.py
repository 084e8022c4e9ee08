"""Pins oracle/pvfitting_oracle.py to the golden vectors the reference itself
produced (tests/golden/make_golden_pv.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import pvfitting_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
AXES = ('major', 'minor')


def gold(name):
    return np.load(os.path.join(GOLD, f'pvfit_{name}.npz'))


def opt(f, key):
    a = f[key]
    return None if a.ndim == 0 and np.isnan(a) else a


def setup_from(f):
    beam = None if np.isnan(f['in_beam']).all() else f['in_beam']
    nnest = opt(f, 'in_nnest')
    lw = opt(f, 'in_linewidth')
    return O.make_setup(
        f['in_x'], f['in_z'], f['in_v'], beam, incl=float(f['in_incl']),
        rc_first=float(f['in_rc_first']),
        nnest=None if nnest is None else [int(n) for n in nnest],
        nsubgrid=int(f['in_nsubgrid']), reslim=float(f['in_reslim']),
        signmajor=int(f['in_signmajor']), signminor=int(f['in_signminor']),
        pa_major=float(f['in_pa_major']), pa_minor=float(f['in_pa_minor']),
        linewidth=None if lw is None else float(lw), rout=float(f['in_rout']))


def test_tables_bitwise():
    f = gold('tables')
    T = O.tables()
    for k in ('Nr', 'Ntheta', 'lnr0', 'dlnr', 'theta0', 'dtheta'):
        assert T[k] == f[k], k
    for k in ('rho_env', 'rho_disk', 'vr_env', 'vt_env', 'vp_all'):
        assert T[k].sum() == f[k + '_sum'], k
        assert np.array_equal(np.abs(T[k]).sum(axis=1), f[k + '_abssum_rows']), k
        assert np.array_equal(T[k][::7, ::13], f[k + '_stride']), k


@pytest.mark.parametrize('name', ['S1', 'S2', 'S3', 'example'])
def test_grid_and_caches_bitwise(name):
    f = gold(name)
    s = setup_from(f)
    g = s['grid']
    assert g['nlevels'] == int(f['grid_nlevels'])
    assert np.array_equal(np.array(g['shape']), f['grid_shapes'])
    assert np.array_equal(np.array([len(c[0]) for c in g['xyz']]), f['grid_ncell'])
    if g['nlevels'] > 1:
        assert np.array_equal(np.array(g['box'][:-1]), f['grid_box'])
    for l in range(g['nlevels']):
        for c, a in zip('xyz', g['axes'][l]):
            assert np.array_equal(a, f[f'grid_axis_{c}{l}']), (l, c)
    assert np.array_equal(s['y'], f['grid_y'])
    assert [s['signmajor'], s['signminor']] == list(f['set_signs'])
    assert s['iradshift'] == f['set_iradshift']
    assert np.array_equal(s['vedge'], f['set_vedge'])
    if 'set_gauss_xy' in f:
        assert np.array_equal(s['gauss_xy'], f['set_gauss_xy'])
    if 'set_gauss_v' in f:
        assert np.array_equal(s['gauss_v'], f['set_gauss_v'])
    full = f'cache_major_0_idx_t' in f
    for a in AXES:
        for l in range(g['nlevels']):
            for k in ('idx_t', 'j_org', 'r_org', 'elos_r', 'elos_t', 'elos_p'):
                mine = s['cache'][a][l][k]
                if full:
                    assert np.array_equal(mine, f[f'cache_{a}_{l}_{k}']), (a, l, k)
                else:
                    assert mine.sum() == f[f'cache_{a}_{l}_{k}_sum'], (a, l, k)
                    assert np.array_equal(mine[::97], f[f'cache_{a}_{l}_{k}_stride'])


@pytest.mark.parametrize('name', ['S1', 'S2', 'S3', 'example'])
def test_mockpvd_matches_reference(name):
    f = gold(name)
    s = setup_from(f)
    for nm in f['vector_names']:
        q = f[f'vec_{nm}_q']
        stages = {}
        major, minor = O.mockpvd(s, *q, stages=stages)
        for mine, key in ((major, 'major'), (minor, 'minor')):
            ref = f[f'vec_{nm}_{key}']
            assert mine.shape == ref.shape
            scale = np.nanmax(np.abs(ref))
            assert np.nanmax(np.abs(mine - ref)) <= 1e-13 * scale, (nm, key)
            assert np.array_equal(np.isnan(mine), np.isnan(ref))


def test_own_tau_bitwise():
    """Per-level column sums (rho2tau outputs) equal the reference's, bit for
    bit, on the small grids."""
    for name in ('S1', 'S2', 'S3'):
        f = gold(name)
        s = setup_from(f)
        g = s['grid']
        L = g['nlevels']
        for nm in f['vector_names']:
            if f'vec_{nm}_own_major_0' not in f:
                continue
            q = f[f'vec_{nm}_q']
            rho, vlos = {}, {}
            for a in AXES:
                pr = [O.rho_vlos(s, a, l, q[0], q[1], q[2], q[4]) for l in range(L)]
                rho[a], vlos[a] = [p[0] for p in pr], [p[1] for p in pr]
            top = max(np.nanmax(r) for a in AXES for r in rho[a])
            for a in AXES:
                r = [x / top for x in rho[a]]
                for l in range(L):
                    shp = g['shape'][l]
                    rl = O.collapse(g, r, upto=l).reshape(shp)
                    vl = O.collapse(g, vlos[a], upto=l).reshape(shp)
                    if g['box'][l] is not None:
                        (x0, x1), (y0, y1), (z0, z1) = g['box'][l]
                        rl[x0:x1 + 1, y0:y1 + 1, z0:z1 + 1] = 0.
                    own = O.rho2tau(s['vedge'], vl, rl)
                    assert np.array_equal(own, f[f'vec_{nm}_own_{a}_{l}']), (name, nm, a, l)


def test_known_answers():
    f = gold('example')
    s = setup_from(f)
    major, minor = O.mockpvd(s, 0.5, 150., 0.6, 10., 10.)
    assert np.nanmax(major) == pytest.approx(0.4066004035049253, rel=1e-13)
    assert np.nansum(major) == pytest.approx(163.9387398678693, rel=1e-13)
    assert np.nansum(minor) == pytest.approx(59.05831187700667, rel=1e-13)


def test_likelihood_matches_reference():
    f = gold('example_fit')
    m = gold('example')
    beam = f['obs_beam']
    s = O.make_setup(f['obs_x'], m['in_z'], f['obs_v'], beam, incl=85.,
                     rc_first=float(f['fit_rc_first']), nnest=[2] * 6,
                     reslim=10, signmajor=-1, signminor=1, pa_major=2.0 - 180,
                     pa_minor=92.0 - 180, rout=float(m['in_rout']))
    obs = O.make_obs(f['obs_x'], f['obs_v'], f['obs_major'], f['obs_minor'],
                     float(f['obs_sigma']), beam[0], beam[1], vmask=(-0.3, 0.3))
    assert [O.getquad(f['obs_major']), O.getquad(f['obs_minor'])] == list(f['obs_getquad'])
    for p, ref in zip(f['fit_P'], f['fit_lnp']):
        # free: log Mstar, log Rc, log taumax, log frho; fixed alphainfall, sig_mdl
        q = np.array([10**p[0], 10**p[1], 0.6, 10**p[2], 10**p[3], 5.0])
        assert O.lnprob(s, obs, q) == pytest.approx(ref, rel=1e-12)
    q = np.array([0.5, 150., 0.6, 10., 10., 5.])
    dof = (f['obs_major'].size + f['obs_minor'].size) / np.sqrt(obs['rarea']) - 7
    assert O.chi2_fixed(s, obs, q) / dof == pytest.approx(
        float(f['fit_chi2r_allfixed']), rel=1e-12)
