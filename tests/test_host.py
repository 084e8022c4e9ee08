"""Host-side logic on a CPU-only box: makegrid constants, the fit driver's
free/fixed/log10 bookkeeping, the sampler, FITS I/O.  Where a likelihood value
is needed the CPU oracle stands in for the GPU engine (tests only)."""
import contextlib
import io
import os

import numpy as np
import pytest

import _golden
from oracle import channelfit_oracle as orc
from slam_amd import fitsio, utils
from slam_amd.channelfit import ChannelFit, vunit


class OracleEngine:
    """Test double with HipLikelihood's host interface, backed by the oracle."""

    def __init__(self, setup):
        self.s = setup
        self.calls = []

    def chi2_batch(self, P):
        P = np.atleast_2d(P)
        self.calls.append(len(P))
        return np.array([orc.chi2_of(self.s, p) for p in P])

    def cubemodel(self, p, convolving=True):
        return orc.cubemodel(self.s, p, convolving=convolving)

    def close(self):
        pass


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def product_fit(g):
    kw = _golden.inputs(g)
    fit = ChannelFit(disk=kw['disk'], envelope=kw['envelope'], progressbar=False)
    fit.set_cube(kw['x'], kw['y'], kw['v'], kw['data'], kw['bmaj'], kw['bmin'],
                 kw['bpa'], kw['sigma'])
    quiet(fit.makegrid, cubefits=None, pa=kw['pa'], incl=kw['incl'], rmax=kw['rmax'],
          vlim=kw['vlim'], nlayer=kw['nlayer'], gaussmargin=kw['gaussmargin'],
          signmajor=kw['signmajor'], signminor=kw['signminor'])
    return fit


@pytest.mark.parametrize('name', _golden.CONFIGS)
def test_makegrid_constants_match_reference(name):
    g = _golden.load(name)
    fit = product_fit(g)
    d = fit.setup_dict()
    for k in ('Xnest0', 'Ynest0', 'gaussbeam', 'xneed', 'yneed', 'v_valid', 'data_valid'):
        np.testing.assert_array_equal(d[k], g['setup_' + k], err_msg=k)
    for k in ('pixperbeam', 'signmajor', 'signminor', 'dv', 'incl0'):
        assert float(d[k]) == float(g['setup_' + k]), k
    for k in ('nq1', 'nq3', 'ineed0', 'ineed1', 'nlayer', 'npix'):
        assert int(d[k]) == int(g['setup_' + k]), k
    assert d['vunit'] == float(g['setup_vunit']) == vunit


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from slam_amd import SlamhipError
    fit = product_fit(_golden.load('F'))
    with pytest.raises(SlamhipError):
        fit.cubemodel(0.5, 150, 0.4)


def test_scaling_variants():
    for sc in ('uniform', 'mom0ft', 'mom0clean', 'mom0model'):
        assert ChannelFit(scaling=sc).scaling == sc
    with pytest.raises(ValueError):
        ChannelFit(scaling='peak')


@pytest.fixture(scope='module')
def fitcase():
    g = _golden.load('A')
    fit = product_fit(g)
    fit.engine = OracleEngine(orc.make_setup(**_golden.inputs(g)))
    return fit, np.load(os.path.join(_golden.GOLDEN_DIR, 'channelfit_fitting.npz'))


FIXED = {'cs': 0.4, 'Rin': 0, 'Ienv': 0, 'xoff': 0, 'yoff': 0, 'voff': 0,
         'incloff': 0, 'paoff': 0}
RANGES = dict(Mstar_range=[0.1, 1.0], Rc_range=[30, 300], pI_range=[0, 3],
              h1_range=[0, 0.5], h2_range=[0, 0.5])


def test_fitting_bookkeeping_matches_reference(fitcase, tmp_path):
    """Prior box, labels and the lnprob closure's values equal what the
    reference's `fitting` hands to its sampler (golden capture)."""
    fit, gold = fitcase
    quiet(fit.fitting, **RANGES, fixed_params=FIXED, filename=str(tmp_path / 'f'),
          kwargs_emcee_corner={'nwalkers_per_ndim': 2, 'nburnin': 2, 'nsteps': 2,
                               'rangelevel': 0.99, 'seed': 3})
    np.testing.assert_array_equal(fit.plim, gold['fit_plim'])
    lnp = fit.lnprob(gold['fit_P'])                    # batched call
    np.testing.assert_allclose(lnp, gold['fit_lnp'], rtol=1e-12)
    one = fit.lnprob(gold['fit_P'][0])                 # scalar call, like the reference
    np.testing.assert_allclose(one, gold['fit_lnp'][0], rtol=1e-12)
    # result containers and the popt.txt layout
    assert list(fit.popt) == fit.paramkeys and set(fit.plow) == set(fit.paramkeys)
    got = open(tmp_path / 'f.popt.txt').read().splitlines()
    ref = str(gold['fit_popt_txt']).splitlines()
    assert got[:2] == ref[:2]
    tab = np.loadtxt(tmp_path / 'f.popt.txt')
    assert tab.shape == (13, 4)
    # fixed parameters come back in natural units
    assert fit.popt['cs'] == 0.4 and fit.popt['Ienv'] == 0.0
    assert 0.1 < fit.popt['Mstar'] < 1.0 and 30 < fit.popt['Rc'] < 300


def test_fitting_batches_half_ensembles(fitcase, tmp_path):
    fit, _ = fitcase
    fit.engine.calls.clear()
    quiet(fit.fitting, **RANGES, fixed_params=FIXED, filename=str(tmp_path / 'g'),
          kwargs_emcee_corner={'nwalkers_per_ndim': 2, 'nburnin': 1, 'nsteps': 1,
                               'seed': 5})
    # ndim = 5 -> 10 walkers: per phase one full-ensemble call + 2 half-ensemble
    # calls per step (rows outside the prior box are never sent)
    assert len(fit.engine.calls) == 6
    assert max(fit.engine.calls) <= 10 and sum(fit.engine.calls) <= 2 * (10 + 10)


def test_all_fixed_gives_reduced_chi2(fitcase, tmp_path):
    fit, gold = fitcase
    allfixed = dict(FIXED, Mstar=0.5, Rc=150., h1=0.1, h2=0.3, pI=1.)
    quiet(fit.fitting, fixed_params=allfixed, filename=str(tmp_path / 'h'))
    np.testing.assert_allclose(fit.chi2r, float(gold['fit_chi2r_allfixed']), rtol=1e-12)
    assert fit.popt['Mstar'] == 0.5


def test_stretch_sampler_recovers_gaussian():
    rng = np.random.default_rng(0)
    mu, sig = np.array([1.0, -2.0, 0.5]), np.array([0.5, 1.5, 0.2])
    seen = []

    def lnp(P):
        seen.append(P.copy())
        return -0.5 * np.sum(((P - mu) / sig)**2, axis=1)

    out, sampler = utils.emcee_corner([mu - 6 * sig, mu + 6 * sig], lnp, nwalkers_per_ndim=16,
                                      nburnin=300, nsteps=600, vectorize=True, seed=1,
                                      simpleoutput=True, return_sampler=True)
    pmid, perr = out
    assert np.all(np.abs(pmid - mu) < 0.15 * sig)
    assert np.all(np.abs(perr / sig - 1) < 0.15)
    assert sampler.chain.shape == (48, 600, 3) and sampler.lnprobability.shape == (48, 600)
    lo, hi = mu - 6 * sig, mu + 6 * sig
    assert all(np.all((lo < P) & (P < hi)) for P in seen)      # strict box prior


def test_emcee_corner_scalar_mode_and_outputs():
    def lnp(p):
        assert np.ndim(p) == 1
        return -0.5 * np.sum(p**2)
    res = utils.emcee_corner([[-5, -5], [5, 5]], lnp, nwalkers_per_ndim=4, nburnin=20,
                             nsteps=40, seed=2, simpleoutput=False)
    popt, plow, pmid, phigh = res
    assert all(len(a) == 2 for a in res) and np.all(plow <= pmid) and np.all(pmid <= phigh)


def test_fits_roundtrip_and_cube_reader(tmp_path):
    nv, ny, nx = 6, 8, 10
    rng = np.random.default_rng(4)
    cube = rng.normal(size=(nv, ny, nx))
    h = {'CTYPE1': 'RA---SIN', 'CRVAL1': 69.9744917, 'CRPIX1': 6.0, 'CDELT1': -1e-5,
         'CTYPE2': 'DEC--SIN', 'CRVAL2': 26.0526194, 'CRPIX2': 5.0, 'CDELT2': 1e-5,
         'CTYPE3': 'FREQ', 'CRVAL3': 2.30538e11, 'CRPIX3': 1.0, 'CDELT3': -1.0e5,
         'CUNIT3': 'Hz', 'RESTFRQ': 2.30538e11, 'BMAJ': 4e-5, 'BMIN': 3e-5, 'BPA': 12.5,
         'BUNIT': 'Jy/beam', 'OBJECT': "it's"}
    path = str(tmp_path / 'cube.fits')
    fitsio.write_cube(path, cube, h)
    d, hh = fitsio.read_primary(path)
    np.testing.assert_array_equal(d, cube)
    assert hh['NAXIS1'] == nx and hh['NAXIS3'] == nv and hh['BPA'] == 12.5
    assert hh['OBJECT'] == "it's" and hh['CUNIT3'] == 'Hz'

    r = utils.ReadFits()
    out = quiet(r.read_cubefits, path, center='04h39m53.878s +26d03m09.43s', dist=140,
                vsys=5.9, sigma=1e-3)
    # offsets of the map centre from `center` (small-angle check of the sky-offset frame)
    ra, dec = utils.parse_center('04h39m53.878s +26d03m09.43s')
    cx = (ra - h['CRVAL1']) * np.cos(np.radians(h['CRVAL2']))
    cy = dec - h['CRVAL2']
    x_ref = ((np.arange(nx) - 5) * h['CDELT1'] - cx) * 3600 * 140
    y_ref = ((np.arange(ny) - 4) * h['CDELT2'] - cy) * 3600 * 140
    np.testing.assert_allclose(out['x'], x_ref, atol=2e-3)
    np.testing.assert_allclose(out['y'], y_ref, atol=2e-3)
    v_ref = (1 - (h['CRVAL3'] + np.arange(nv) * h['CDELT3']) / h['RESTFRQ']) * 299792.458 - 5.9
    np.testing.assert_allclose(out['v'], v_ref, rtol=1e-12)
    assert r.bmaj == pytest.approx(4e-5 * 3600 * 140) and r.bpa == 12.5
    assert r.dv == pytest.approx(v_ref[1] - v_ref[0])
    # cropping keeps the pixels nearest to the limits (utils.py:249-258)
    out = quiet(r.read_cubefits, path, dist=140, xmin=-10, xmax=10, ymin=-10, ymax=10, sigma=1e-3)
    assert out['data'].shape[0] == nv and out['x'].max() <= 12 and out['x'].min() >= -12


def test_spherical_offsets_hand_computed():
    """The sky-offset frame of `read_cubefits` (reference utils.py:211-216 asks
    astropy's `spherical_offsets_to`; astropy is not importable here, so the
    function is pinned on cases worked by hand from the frame's definition:
    lon = atan2(p.e, p.o), lat = asin(p.n) with o the origin's unit vector and
    e, n the local east and north there)."""
    from slam_amd.utils import parse_center, spherical_offsets
    sec = 1.0 / 3600.0
    # the origin itself; a pure declination offset; an offset along the equator
    assert spherical_offsets(60.0, 30.0, 60.0, 30.0) == (0.0, 0.0)
    lon, lat = spherical_offsets(60.0, 30.0, 60.0, 31.0)
    assert abs(lon) < 1e-13 and abs(lat - 1.0) < 1e-13
    lon, lat = spherical_offsets(10.0, 0.0, 12.5, 0.0)
    assert abs(lon - 2.5) < 1e-13 and abs(lat) < 1e-13
    # o = (cos45, 0, sin45), p = (0, 1, 0): e = (0, 1, 0), n = (-sin45, 0, cos45) -> (90, 0)
    lon, lat = spherical_offsets(0.0, 45.0, 90.0, 0.0)
    assert abs(lon - 90.0) < 1e-13 and abs(lat) < 1e-13
    # over the pole: 20 degrees due north of (ra0, 80) lies (ra0 + 180, 80)
    lon, lat = spherical_offsets(33.0, 80.0, 213.0, 80.0)
    assert abs(lon) < 1e-12 and abs(lat - 20.0) < 1e-12
    # small angles at declination 60: lon = dRA cos(dec), lat = dRA^2 sin(dec) cos(dec) / 2 (the
    # parallel lies poleward of the great circle), both to second order
    dra = 1.0 * sec
    lon, lat = spherical_offsets(70.0, 60.0, 70.0 + dra, 60.0)
    assert abs(lon / (dra * 0.5) - 1) < 1e-9
    expect = np.degrees(np.radians(dra)**2 * np.sin(np.radians(60)) * 0.5 / 2)
    assert abs(lat / expect - 1) < 1e-4          # 3e-10 deg out of asin(~1e-11): rounding is 1e-6 of it
    # west of the origin is negative longitude (R.A. decreases to the west)
    assert spherical_offsets(70.0, 60.0, 70.0 - dra, 60.0)[0] < 0
    ra, dec = parse_center('04h39m53.878s +26d03m09.43s')
    assert abs(ra - (4 + 39 / 60 + 53.878 / 3600) * 15) < 1e-12
    assert abs(dec - (26 + 3 / 60 + 9.43 / 3600)) < 1e-12
    assert parse_center('16h00m00s -00d30m00s')[1] == -0.5


def test_fitspace_bookkeeping(tmp_path, capsys):
    """Free / fixed / log-sampled rules shared by ChannelFit.fitting and fit_mockpvd."""
    from slam_amd.fitspace import FitSpace
    keys = ['a', 'b', 'c', 'd']
    sp = FitSpace(keys, [0, 1], {'b': 100., 'd': 3.}, [[0.1, 10], [1, 1000], [-2, 2], [0, 5]])
    assert sp.free == {'a': True, 'b': False, 'c': True, 'd': False} and sp.ndim == 2
    assert np.array_equal(sp.plim, [[-1., -2.], [1., 2.]])            # log10 box of a, plain box of c
    assert sp.base[1] == 2.0 and sp.base[3] == 3.0                    # fixed b held as log10
    assert np.array_equal(sp.to_natural([0.5, 1.5]), [[10**0.5, 100., 1.5, 3.]])
    assert sp.to_natural(np.zeros((7, 2))).shape == (7, 4)
    assert list(sp.labels(keys)) == ['loga', 'c']
    assert sp.corner_ranges(None) is None
    r = sp.corner_ranges({'a': [1., 10.], 'c': 0.9})
    assert np.array_equal(r[0], [0., 1.]) and r[1] == 0.9
    assert sp.corner_ranges({'b': [1, 2]}) == [0.8, 0.8]              # only free parameters are listed
    allfixed = FitSpace(keys, [0, 1], dict(zip(keys, [2., 100., 0., 3.])), [[1, 2]] * 4)
    assert not allfixed.any_free and np.array_equal(allfixed.base, [2., 100., 0., 3.])   # natural
    sp.report([1, 2, 3, 4], [5, 6, 7, 8], [9, 10, 11, 12], [2, 0, 1, 2], ['x', 'y', '', 'z'],
              ['f', 'f', 'e', 'f'])
    assert capsys.readouterr().out.splitlines()[2] == 'c = 3.0e+00, 7.0e+00, 1.1e+01 '
    path = tmp_path / 't.popt.txt'
    sp.write_table(path, [1, 2, 3, 4], [0, 0, 0, 0], [1, 1, 1, 1], [2, 2, 2, 2])
    txt = path.read_text().splitlines()
    assert txt[0] == '#Rows:a,b,c,d' and txt[1] == '#Columns:popt,plow,pmid,phigh'
    assert np.array_equal(np.loadtxt(path)[:, 0], [1, 2, 3, 4])


class _StubResults:
    def __init__(self, samples, logl, logwt):
        self.samples, self.logl, self._logwt = samples, logl, logwt
        self.logz, self.logzerr = np.array([-3.0, -2.5]), np.array([0.2, 0.1])

    def importance_weights(self):
        w = np.exp(self._logwt - self._logwt.max())
        return w / w.sum()


class _StubNested:
    """Stands in for dynesty.DynamicNestedSampler (not importable here): same
    constructor and run_nested/results surface; evaluates `nlive` prior draws
    through the pool when one is given (as dynesty does for its live points),
    then a few single points through the scalar likelihood."""
    last = None

    def __init__(self, loglikelihood, prior_transform, ndim, pool=None, queue_size=None):
        self.ll, self.pt, self.ndim, self.pool, self.queue_size = (
            loglikelihood, prior_transform, ndim, pool, queue_size)
        _StubNested.last = self

    def run_nested(self, print_progress=True, wt_kwargs=None):
        assert print_progress is False
        rng = np.random.default_rng(7)
        u = rng.random((64, self.ndim))
        v = np.array([self.pt(x) for x in u])
        if self.pool is not None:
            logl = np.array(self.pool.map(self.ll, v))
        else:
            logl = np.array([self.ll(x) for x in v])
        extra = np.array([self.pt(x) for x in rng.random((5, self.ndim))])
        logl = np.append(logl, [self.ll(x) for x in extra])
        v = np.vstack([v, extra])
        self.results = _StubResults(v, logl, logl.copy())


def test_dynesty_corner_hook_with_stub_sampler(capsys):
    """utils.py:122-174's driver: prior transform onto the box, the vectorised
    likelihood reached through the BatchedPool for whole queues and through one-row
    batches otherwise, reference return values.  dynesty itself is absent here:
    PARITY UNPINNED for real nested-sampling runs."""
    bounds = np.array([[0.0, -1.0], [2.0, 3.0]])
    mu = np.array([1.2, 0.5])
    batches = []

    def lnp_vec(P, scale):
        P = np.asarray(P)
        assert P.ndim == 2
        batches.append(len(P))
        return -0.5 * np.sum(((P - mu) / scale)**2, axis=1)

    out = utils.dynesty_corner(bounds, lnp_vec, args=[0.1], vectorize=True, simpleoutput=False,
                               return_evidence=True, sampler_cls=_StubNested)
    popt, plow, pmid, phigh = out
    st = _StubNested.last
    assert st.queue_size == 16 and st.pool.size == 16            # 8 x ndim
    assert batches[0] == 64 and batches[1:] == [1] * 5            # live points as ONE call
    assert st.pool.batched_calls == 1
    assert np.all(bounds[0] <= st.results.samples) and np.all(st.results.samples <= bounds[1])
    assert np.array_equal(popt, st.results.samples[np.argmax(st.results.logl)])
    assert np.all(plow <= pmid) and np.all(pmid <= phigh)
    assert np.allclose(pmid, mu, atol=0.15)
    assert 'Evidence:' in capsys.readouterr().out
    # scalar mode, simple output; a pool asked to map something else maps it serially
    pmid2, perr2 = utils.dynesty_corner(bounds, lambda p, s: float(lnp_vec(p[None], s)[0]), args=[0.1],
                                        sampler_cls=_StubNested)
    assert _StubNested.last.pool is None
    assert np.allclose(pmid2, pmid) and np.allclose(perr2, (phigh - plow) / 2)
    assert st.pool.map(lambda x: x + 1, [1, 2]) == [2, 3] and st.pool.serial_calls == 1


def test_dynesty_corner_without_dynesty_says_so():
    try:
        import dynesty  # noqa: F401
        pytest.skip('dynesty is installed')
    except ImportError:
        pass
    with pytest.raises(ImportError, match='dynesty'):
        utils.dynesty_corner([[0.], [1.]], lambda p: 0.0)


def test_weighted_quantile_rule():
    x = np.array([3., 1., 2., 4.])
    w = np.array([0.1, 0.4, 0.4, 0.1])
    # sorted 1,2,3,4 with preceding cumulative weights 0, .4, .8, .9 -> /0.9
    q = utils.weighted_quantile(x, [0.0, 0.4 / 0.9, 0.5, 1.0], weights=w)
    assert q[0] == 1.0 and q[1] == pytest.approx(2.0) and q[3] == 4.0
    assert q[2] == pytest.approx(2.0 + (0.5 - 0.4 / 0.9) / (0.4 / 0.9))
    assert utils.weighted_quantile(x, [0.5]) == [2.5]


def test_bench_quotes_measured_traffic_only_for_the_build_it_was_measured_on():
    """bench.py's roofline.traffic comes from the committed PMC profile: with the library's
    sources unchanged since the profile it is a number for the beam kernel (whatever instance
    name the kernel has in the profile), otherwise None with the reason -- never stale bytes."""
    import json
    import os
    import bench
    from slam_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = json.load(open(os.path.join(root, 'profiles', bench.TRAFFIC_PROFILE)))
    meta = prof['_meta']

    class Args:
        config, bpa, scaling, workload, dtype = (meta['config'], meta['bpa'], meta['scaling'],
                                                 meta['workload'], meta['dtype'])
        walkers = 8 * meta['walkers_per_launch']

    class Eng:
        walkers_per_group = meta['walkers_per_launch']

    got, why = bench.measured_traffic('k_conv', Args, Eng)
    if meta['src_sha'] == build.src_sha('slamhip'):
        assert got is not None and 1e9 < got < 1e11, why
        assert bench.measured_traffic('k_profile', Args, Eng)[0] is not None
    else:
        assert got is None and 'stale profile' in why
    Args.workload = 'dense_disk'
    assert bench.measured_traffic('k_conv', Args, Eng)[0] is None
