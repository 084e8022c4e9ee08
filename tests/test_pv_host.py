"""CPU-only checks of the mock-PV product's host side: the once-per-fit constants
equal the reference's caches bit for bit, the C-ABI library exports what
include/slampv.h declares, FITS reading, and that nothing runs without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import pvfitting_oracle as O
from slam_amd import _pvlib, fitsio, pvgrid
from slam_amd._lib import SlamhipError
from slam_amd.mockpvd import MockPVD
from slam_amd.pvfitting import PVFitting, getquad

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
AXES = ('major', 'minor')


def gold(name):
    return np.load(os.path.join(GOLD, f'pvfit_{name}.npz'))


def opt(f, key):
    a = f[key]
    return None if a.ndim == 0 and np.isnan(a) else a


def model_from(f):
    beam = None if np.isnan(f['in_beam']).all() else f['in_beam']
    nnest = opt(f, 'in_nnest')
    return MockPVD(f['in_x'], f['in_z'], f['in_v'],
                   nnest=None if nnest is None else [int(n) for n in nnest],
                   nsubgrid=int(f['in_nsubgrid']), beam=beam,
                   reslim=float(f['in_reslim']), signmajor=int(f['in_signmajor']),
                   signminor=int(f['in_signminor']), pa_major=float(f['in_pa_major']),
                   pa_minor=float(f['in_pa_minor']))


# ---------------------------------------------------------------------------
def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'slampv.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(slampv_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported_and_bound():
    names = declared_symbols()
    assert 'slampv_lnprob_batch' in names and 'slampv_create' in names
    lib = ctypes.CDLL(_pvlib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f'{n} declared in slampv.h but not exported'
    assert sorted(_pvlib.SYMBOLS) == names


def _struct_fields(name):
    text = open(os.path.join(ROOT, 'include', 'slampv.h')).read()
    body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (name, name), text, re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split('*', 1)[1] if '*' in decl else decl.split(None, 1)[1]
        fields += [re.sub(r'\[.*?\]', '', n).strip() for n in names.split(',')]
    return fields


def test_struct_layouts_match_header():
    assert _struct_fields('slampv_level') == [f[0] for f in _pvlib.SlampvLevel._fields_]
    assert _struct_fields('slampv_desc') == [f[0] for f in _pvlib.SlampvDesc._fields_]


def test_library_reports_itself():
    lib = _pvlib.load()
    assert lib.slampv_version().decode().startswith('slampv')
    assert lib.slampv_kernel_count() == 6
    assert lib.slampv_create(None, None) < 0


# ---------------------------------------------------------------------------
def test_tables_bitwise():
    f = gold('tables')
    T = pvgrid.tables()
    assert (T.nr, T.ntheta) == (int(f['Nr']), int(f['Ntheta']))
    for k in ('lnr0', 'dlnr', 'theta0', 'dtheta'):
        assert getattr(T, k) == f[k], k
    for k in ('rho_env', 'rho_disk', 'vr_env', 'vt_env', 'vp_all'):
        a = getattr(T, k)
        assert a.sum() == f[k + '_sum'], k
        assert np.array_equal(a[::7, ::13], f[k + '_stride']), k
    assert T.packed().shape == (T.ntheta, T.nr, _pvlib.NTAB)
    assert np.isfinite(T.packed()).all()


@pytest.mark.parametrize('name', ['S1', 'S2', 'S3', 'example'])
def test_grid_and_cell_constants_bitwise(name):
    """Product setup == the reference's caches, cell for cell."""
    f = gold(name)
    m = model_from(f)
    assert np.array_equal(np.array([L.shape for L in m.levels]), f['grid_shapes'])
    assert np.array_equal(m.y, f['grid_y'])
    assert [m.signmajor, m.signminor] == list(f['set_signs'])
    assert m.iradshift == f['set_iradshift']
    for l, L in enumerate(m.levels):
        for c, a in zip('xyz', (L.x, L.y, L.z)):
            assert np.array_equal(a, f[f'grid_axis_{c}{l}'])
    if len(m.levels) > 1:
        assert np.array_equal(np.array([L.box for L in m.levels[:-1]]), f['grid_box'])
    lw = opt(f, 'in_linewidth')
    m._freeze(float(f['in_rc_first']), float(f['in_incl']),
              None if lw is None else float(lw), m.pa_major)
    fz = m._frozen
    if 'set_gauss_xy' in f:
        assert np.array_equal(fz['gauss_xy'], f['set_gauss_xy'][0])
    if 'set_gauss_v' in f:
        assert np.array_equal(fz['gauss_v'], f['set_gauss_v'][:, 0, 0])
    assert np.array_equal(pvgrid.velocity_edges(m.v), f['set_vedge'])
    # the oracle's list order maps the full boxes onto the reference's lists
    og = O.nested_grid(m.x, m.y, m.z, list(m.nnest or []), float(f['in_reslim']))
    full = f'cache_major_0_idx_t' in f
    for a in AXES:
        for l, L in enumerate(m.levels):
            perm = og['perm'][l]
            for k in ('idx_t', 'j_org', 'r_org', 'elos_r', 'elos_t', 'elos_p'):
                mine = m._consts[a][l][k].ravel()
                if perm is not None:
                    mine = mine[perm]
                if full:
                    assert np.array_equal(mine, f[f'cache_{a}_{l}_{k}']), (a, l, k)
                else:
                    assert mine.sum() == f[f'cache_{a}_{l}_{k}_sum'], (a, l, k)
                    assert np.array_equal(mine[::97], f[f'cache_{a}_{l}_{k}_stride'])


def test_device_layout_flags():
    f = gold('S1')
    m = model_from(f)
    m._freeze(60., 60., 0.5, m.pa_major)
    L = m.levels[0]
    lay = pvgrid.device_layout(L, m._consts['major'][0], rin=1., rout=float(f['in_rout']))
    nx, ny, nz = L.shape
    assert lay['idx_t'].shape == (nz, ny, nx) and lay['idx_t'].dtype == np.int32
    r = m._consts['major'][0]['r_org'].transpose(2, 1, 0)
    cov = L.covered().transpose(2, 1, 0)
    dead = lay['idx_t'] < 0
    assert np.array_equal(dead, cov | (r <= 1.) | (r > float(f['in_rout'])))
    assert cov.sum() == np.prod([b[1] - b[0] + 1 for b in L.box])
    assert (lay['idx_t'][~dead] < pvgrid.tables().ntheta).all()


def test_getquad_and_obs():
    f = gold('example_fit')
    assert [getquad(f['obs_major']), getquad(f['obs_minor'])] == list(f['obs_getquad'])


def test_read_pvfits_roundtrip(tmp_path):
    """read_pvfits on a PV FITS written with our writer: axes, cropping,
    skipping and beam as utils.py:279-352 derives them."""
    nx, nv = 41, 24
    rng = np.random.default_rng(3)
    d = rng.normal(size=(1, nv, nx)).astype(np.float64)
    h = {'CRPIX1': 21.0, 'CDELT1': 0.05, 'CRVAL1': 0.0, 'CTYPE1': 'OFFSET',
         'CRPIX2': 1.0, 'CDELT2': 200.0, 'CRVAL2': -2000.0, 'CUNIT2': 'm/s',
         'BMAJ': 0.3 / 3600, 'BMIN': 0.2 / 3600, 'BPA': 12.0}
    path = str(tmp_path / 'pv.fits')
    fitsio.write_cube(path, d, h)
    pv = PVFitting()
    out = pv.read_pvfits(path, dist=100., vsys=0.5, xmin=-60, xmax=60, vmin=-1.9,
                         vmax=1.5, xskip=2, sigma=0.1)
    xs = ((np.arange(nx) - 20) * 0.05)[0::2] * 100.
    i0, i1 = np.argmin(np.abs(xs + 60)), np.argmin(np.abs(xs - 60))
    vs = (np.arange(nv) * 200. - 2000.) * 1e-3 - 0.5
    k0, k1 = np.argmin(np.abs(vs + 1.9)), np.argmin(np.abs(vs - 1.5))
    assert np.allclose(out['x'], xs[i0:i1 + 1]) and np.allclose(out['v'], vs[k0:k1 + 1])
    assert np.array_equal(out['data'], d[0][k0:k1 + 1, 0::2][:, i0:i1 + 1])
    assert pv.offpix == (i0, k0)
    assert np.allclose(pv.beam, [30., 20., 12.])
    assert pv.header['CRPIX1'] == 11 and pv.header['CDELT1'] == pytest.approx(0.1)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    f = gold('S3')
    m = model_from(f)
    with pytest.raises(SlamhipError):
        m.generate_mockpvd(0.6, 70., 0.5, 2., 5., incl=45., rout=float(f['in_rout']))


def test_modeltofits_writes_padded_diagrams(tmp_path, capsys):
    """PVFitting.modeltofits (_pvfitting.py:426-459): four files, NaN rows for the
    velocity channels cropped away, CRPIX1 shifted by the cropped columns, and the
    stored header untouched (the reference edits it in place)."""
    nx, nv = 41, 24
    d = np.random.default_rng(4).normal(size=(1, nv, nx))
    h = {'CRPIX1': 21.0, 'CDELT1': 0.05, 'CRVAL1': 0.0, 'CTYPE1': 'OFFSET',
         'CRPIX2': 1.0, 'CDELT2': 200.0, 'CRVAL2': -2000.0, 'CUNIT2': 'm/s',
         'BMAJ': 0.3 / 3600, 'BMIN': 0.2 / 3600, 'BPA': 12.0}
    path = str(tmp_path / 'pv.fits')
    fitsio.write_cube(path, d, h)
    pv = PVFitting()
    pv.put_PV(path, path, dist=100., vsys=0.5, rmax=60., vmin=-1.9, vmax=1.5, sigma=0.1)
    n_v, n_x = pv.dpvmajor.shape
    assert len(pv.vorg) == nv and n_v < nv
    model = np.arange(n_v * n_x, dtype=float).reshape(n_v, n_x)
    seen = {}
    pv.makemodel = lambda **p: (seen.update(p) or (model, 2 * model))
    pv.popt = dict(Mstar=1., Rc=100., alphainfall=1., taumax=1., frho=1., sig_mdl=0.3)
    crpix = pv.header['CRPIX1']
    head = str(tmp_path / 'best')
    pv.modeltofits(filehead=head)
    pv.modeltofits(filehead=head)                      # twice: no drift of CRPIX1
    assert 'sig_mdl' not in seen and pv.header['CRPIX1'] == crpix
    nblue = int(np.sum(pv.vorg < pv.v.min()))
    nred = int(np.sum(pv.vorg > pv.v.max()))
    assert nblue + nred + n_v == nv and nblue > 0 and nred > 0
    for ext, want in (('model.major', model), ('model.minor', 2 * model),
                      ('residual.major', pv.dpvmajor - model),
                      ('residual.minor', pv.dpvminor - 2 * model)):
        got, hh = fitsio.read_primary(f'{head}.{ext}.fits')
        assert got.shape == (nv, n_x)
        assert np.isnan(got[:nblue]).all() and np.isnan(got[nblue + n_v:]).all()
        assert np.array_equal(got[nblue:nblue + n_v], want)
        assert hh['CRPIX1'] == crpix - pv.offpix[0] and hh['NAXIS1'] == n_x
    pv.modeltofits(filehead=head, Mstar=2., Rc=50., alphainfall=1., taumax=1., frho=1.)
    assert seen['Mstar'] == 2.
