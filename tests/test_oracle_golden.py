"""Pin the CPU oracle (oracle/channelfit_oracle.py) to outputs of the
reference itself (tests/golden/*.npz, written by tests/golden/make_golden.py)."""
import numpy as np
import pytest

import _golden
from oracle import channelfit_oracle as orc


@pytest.fixture(scope='module', params=_golden.CONFIGS)
def case(request):
    g = _golden.load(request.param)
    return g, orc.make_setup(**_golden.inputs(g))


def test_vunit():
    assert orc.VUNIT == 29.784691829676934


def test_setup_constants(case):
    g, s = case
    for k in ('Xnest0', 'Ynest0', 'gaussbeam', 'xneed', 'yneed', 'v_valid',
              'data_valid'):
        np.testing.assert_array_equal(s[k], g['setup_' + k], err_msg=k)
    for k in ('pixperbeam', 'signmajor', 'signminor', 'dv', 'incl0'):
        assert s[k] == float(g['setup_' + k]), k
    for k in ('nq1', 'nq3', 'ineed0', 'ineed1', 'nlayer', 'npix'):
        assert s[k] == int(g['setup_' + k]), k


def test_stages_and_chi2(case):
    g, s = case
    for name in _golden.vectors(g):
        pre = f'vec_{name}_'
        st = {}
        model = orc.cubemodel(s, g[pre + 'p'], stages=st)
        np.testing.assert_array_equal(st['prof'], g[pre + 'prof'], err_msg=name)
        assert st['prof_d'] == float(g[pre + 'prof_d'])
        np.testing.assert_array_equal(st['Iunif'].sum(axis=(1, 2)),
                                      g[pre + 'Iunif_chsum'], err_msg=name)
        np.testing.assert_allclose(st['Iconv'].sum(axis=(1, 2)),
                                   g[pre + 'Iconv_chsum'], rtol=1e-12,
                                   atol=1e-12)
        np.testing.assert_allclose(st['gf'], float(g[pre + 'gf']), rtol=1e-12)
        np.testing.assert_allclose(st['ff'], float(g[pre + 'ff']), rtol=1e-12)
        np.testing.assert_allclose(st['scale'], float(g[pre + 'scale']),
                                   rtol=1e-12)
        np.testing.assert_allclose(orc.chi2(s, model), float(g[pre + 'chi2']),
                                   rtol=1e-12)
        if pre + 'model' in g:
            np.testing.assert_array_equal(st['Iunif'], g[pre + 'Iunif'])
            assert _golden.relmax(st['Iconv'], g[pre + 'Iconv']) < 1e-13
            assert _golden.relmax(model, g[pre + 'model']) < 1e-13


def test_geometry_taps(case):
    g, s = case
    name = _golden.vectors(g)[0]
    st = {}
    orc.cubemodel(s, g[f'vec_{name}_p'], stages=st)
    for k in range(4):
        if f'geom_xdisk{k}' in g:
            np.testing.assert_array_equal(st['xdisk'][k], g[f'geom_xdisk{k}'])
            np.testing.assert_array_equal(st['vlos'][k], g[f'geom_vlos{k}'])
        else:
            assert st['xdisk'][k] is None


def test_noconv(case):
    g, s = case
    name = _golden.vectors(g)[0]
    m = orc.cubemodel(s, g[f'vec_{name}_p'], convolving=False)
    assert _golden.relmax(m, g['noconv_model']) < 1e-13
    np.testing.assert_allclose(orc.chi2(s, m), float(g['noconv_chi2']),
                               rtol=1e-12)


def test_known_answer():
    """chi2 = 4229.974367626359 on the reference."""
    k = np.load(_golden.GOLDEN_DIR + '/channelfit_known.npz')
    n, nv = 64, 32
    dpix = 400 / 64
    data = np.random.default_rng(0).normal(0, 1.7e-3, (nv, n, n))
    s = orc.make_setup(-(np.arange(n) - 32) * dpix, (np.arange(n) - 32) * dpix,
                       (np.arange(nv) - 16 + 0.5) * 0.25, data, bmaj=6 * dpix,
                       bmin=4 * dpix, bpa=20., sigma=1.7e-3, pa=2., incl=85.,
                       rmax=190., vlim=(-4, -0.5, 0.5, 4), signmajor=1,
                       signminor=1)
    p = [0.5, 150, 0.4, 0.1, 0.3, 1, 0, 0, 0, 0, 0, 0, 0]
    c = orc.chi2_of(s, p)
    assert float(k['chi2']) == 4229.974367626359
    np.testing.assert_allclose(c, 4229.974367626359, rtol=1e-13)


def test_channel_chunks_change_nothing(case):
    """`kchunk` (the oracle evaluating the cube a few channels at a time, which is how the
    full-size config-4 GPU test affords it) gives the unchunked model bit for bit."""
    g, s = case
    name = _golden.vectors(g)[0]
    p = g[f'vec_{name}_p']
    whole = orc.cubemodel(s, p)
    for kc in (1, 3):
        np.testing.assert_array_equal(orc.cubemodel(s, p, kchunk=kc), whole)
    assert orc.chi2_of(s, p, kchunk=2) == orc.chi2_of(s, p)
