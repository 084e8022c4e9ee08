"""GPU tests of the product path (`slam_amd.ChannelFit` -> C ABI -> HIP) and
size-independent properties at BASELINE.json's full sizes, where the oracle is
too slow to sweep."""
import contextlib
import io
import os

import numpy as np
import pytest

import _golden
from oracle import channelfit_oracle as orc

pytestmark = pytest.mark.gpu


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def product_fit(g):
    from slam_amd.channelfit import ChannelFit
    kw = _golden.inputs(g)
    fit = ChannelFit(disk=kw['disk'], envelope=kw['envelope'], progressbar=False)
    fit.set_cube(kw['x'], kw['y'], kw['v'], kw['data'], kw['bmaj'], kw['bmin'],
                 kw['bpa'], kw['sigma'])
    quiet(fit.makegrid, cubefits=None, pa=kw['pa'], incl=kw['incl'], rmax=kw['rmax'],
          vlim=kw['vlim'], nlayer=kw['nlayer'], gaussmargin=kw['gaussmargin'],
          signmajor=kw['signmajor'], signminor=kw['signminor'])
    return fit


def test_channelfit_cubemodel_matches_reference():
    g = _golden.load('A')
    fit = product_fit(g)
    for name in ('generic', 'all_off', 'xyoff_big'):
        p = dict(zip(fit.paramkeys, g[f'vec_{name}_p']))
        m = fit.cubemodel(**p)
        assert _golden.relmax(m, g[f'vec_{name}_model']) < 1e-9
    m0 = fit.cubemodel(**dict(zip(fit.paramkeys, g['vec_generic_p'])), convolving=False)
    assert _golden.relmax(m0, g['noconv_model']) < 1e-9
    fit.close()


def test_channelfit_fitting_lnprob_matches_reference(tmp_path):
    """lnprob handed to the sampler == the reference closure's values."""
    g = _golden.load('A')
    gold = np.load(os.path.join(_golden.GOLDEN_DIR, 'channelfit_fitting.npz'))
    fit = product_fit(g)
    fixed = {'cs': 0.4, 'Rin': 0, 'Ienv': 0, 'xoff': 0, 'yoff': 0, 'voff': 0,
             'incloff': 0, 'paoff': 0}
    quiet(fit.fitting, Mstar_range=[0.1, 1.0], Rc_range=[30, 300], pI_range=[0, 3],
          h1_range=[0, 0.5], h2_range=[0, 0.5], fixed_params=fixed,
          filename=str(tmp_path / 'f'),
          kwargs_emcee_corner={'nwalkers_per_ndim': 4, 'nburnin': 20, 'nsteps': 20, 'seed': 1})
    np.testing.assert_allclose(fit.lnprob(gold['fit_P']), gold['fit_lnp'], rtol=1e-9)
    # the short chain must have moved towards the truth (Mstar=0.5, Rc=150) region
    assert np.isfinite(fit.popt['Mstar']) and os.path.exists(tmp_path / 'f.popt.txt')
    allfixed = dict(fixed, Mstar=0.5, Rc=150., h1=0.1, h2=0.3, pI=1.)
    quiet(fit.fitting, fixed_params=allfixed, filename=str(tmp_path / 'h'))
    np.testing.assert_allclose(fit.chi2r, float(gold['fit_chi2r_allfixed']), rtol=1e-9)
    fit.close()


def test_torch_device_entry_point():
    import torch
    g = _golden.load('B')
    fit = product_fit(g)
    names = _golden.vectors(g)
    P = np.array([g[f'vec_{n}_p'] for n in names])
    ref = np.array([float(g[f'vec_{n}_chi2']) for n in names])
    out = fit.engine.chi2_batch_torch(torch.as_tensor(P, device='cuda'))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-9)
    assert np.array_equal(out.cpu().numpy(), fit.engine.chi2_batch(P))
    fit.close()


def test_nan_data_follows_reference():
    """A NaN in the cube makes np.sum() NaN -> scale NaN -> model NaN ->
    nansum(...) = 0 in the reference (:486-489, :601)."""
    from slam_amd import HipLikelihood
    g = _golden.load('F')
    kw = _golden.inputs(g)
    kw['data'] = kw['data'].copy()
    kw['data'][3, 5, 7] = np.nan
    s = orc.make_setup(**kw)
    p = g['vec_deep_p']
    assert orc.chi2_of(s, p) == 0.0
    d = dict(s)
    d['vunit'] = orc.VUNIT
    with HipLikelihood(d) as eng:
        assert eng.chi2_batch(p[None])[0] == 0.0


def test_negative_scale_clamps_to_zero():
    """gf < 0 -> scale 0 -> chi2 = sum(d^2)/sigma^2/pixperbeam (:489)."""
    from slam_amd import HipLikelihood
    g = _golden.load('F')
    kw = _golden.inputs(g)
    kw['data'] = -np.abs(kw['data'])
    s = orc.make_setup(**kw)
    d = dict(s)
    d['vunit'] = orc.VUNIT
    p = g['vec_deep_p']
    with HipLikelihood(d) as eng:
        st = eng.chi2_stats(p[None])
    assert st['gf'][0] < 0 and st['scale'][0] == 0.0
    expect = np.sum(kw['data']**2) / s['sigma']**2 / s['pixperbeam']
    np.testing.assert_allclose(st['chi2'][0], expect, rtol=1e-12)
    np.testing.assert_allclose(st['chi2'][0], orc.chi2_of(s, p), rtol=1e-12)


@pytest.fixture(scope='module')
def c2():
    from slam_amd import synth
    fit = synth.make_fit('C2')
    yield fit
    fit.close()


def test_c2_vs_oracle(c2):
    """BASELINE config 2 (256x256x64, fp64 parity): 4 of the 64 walkers against
    the oracle (0.8 s each), all 64 for internal consistency."""
    from slam_amd import synth
    P = synth.walkers(64)
    got = c2.chi2_batch(P)
    s = orc.make_setup(c2.x, c2.y, c2.v, c2.data, bmaj=c2.bmaj, bmin=c2.bmin, bpa=c2.bpa,
                       sigma=c2.sigma, pa=2.0, incl=60.0, rmax=synth.RMAX, signmajor=1,
                       signminor=1)
    for i in (0, 17, 40, 63):
        st = {}
        model = orc.cubemodel(s, P[i], stages=st)
        np.testing.assert_allclose(got[i], orc.chi2(s, model), rtol=1e-9)
        if i == 0:
            assert _golden.relmax(c2.engine.iunif(P[i]), st['Iunif']) < 1e-12
            assert _golden.relmax(c2.engine.cubemodel(P[i]), model) < 1e-9
    # fused hot path vs the unfused cubemodel path, every walker
    for i in range(0, 64, 7):
        _, st = c2.engine.cubemodel(P[i], return_stats=True)
        np.testing.assert_allclose(got[i], st['chi2'], rtol=1e-11)


def test_c2_rotated_beam_direct2d():
    from slam_amd import synth
    fit = synth.make_fit('C2', bpa=20.0)
    assert not fit.engine.separable_beam
    P = synth.walkers(3)
    got = fit.chi2_batch(P)
    s = orc.make_setup(fit.x, fit.y, fit.v, fit.data, bmaj=fit.bmaj, bmin=fit.bmin, bpa=fit.bpa,
                       sigma=fit.sigma, pa=2.0, incl=60.0, rmax=synth.RMAX, signmajor=1,
                       signminor=1)
    np.testing.assert_allclose(got[0], orc.chi2_of(s, P[0]), rtol=1e-9)
    # mode 0 took direct taps in the fused kernel (a rotated Gaussian needs ~14
    # outer products); the factorised beam (2) and the dense taps (1) agree
    assert fit.engine.beam_rank == 0
    fit.engine.set_conv_mode(2)
    np.testing.assert_allclose(fit.chi2_batch(P), got, rtol=1e-11)
    fit.engine.set_conv_mode(1)
    np.testing.assert_allclose(fit.chi2_batch(P), got, rtol=1e-11)
    fit.engine.set_conv_mode(0)
    assert np.array_equal(fit.chi2_batch(P[::-1])[::-1], got)
    fit.close()


def test_c2_rotated_beam_truncated_factorisation():
    """Opt-in `beam_tol` (slamhip_desc::beam_tol; no counterpart in the reference): a rotated
    beam as its leading singular triplets.  Tolerance of THIS mode, written here: 1e-5 relative
    on chi2 (north_star's fp64 bound); the default (beam_tol = 0) stays exact and is what every
    other test checks at 1e-9."""
    from slam_amd import synth
    exact = synth.make_fit('C2', bpa=20.0)
    P = synth.walkers(64)
    ref = exact.chi2_batch(P)
    s = orc.make_setup(exact.x, exact.y, exact.v, exact.data, bmaj=exact.bmaj, bmin=exact.bmin,
                       bpa=exact.bpa, sigma=exact.sigma, pa=2.0, incl=60.0, rmax=synth.RMAX,
                       signmajor=1, signminor=1)
    want = np.array([orc.chi2_of(s, P[i]) for i in (0, 33)])
    exact.close()
    for tol, max_rank in ((1e-5, 5), (1e-7, 7)):
        fit = synth.make_fit('C2', bpa=20.0, beam_tol=tol)
        eng = fit.engine
        if eng.beam_rank == 0:            # more terms than pay off: direct taps, i.e. exact
            eng.set_conv_mode(2)          # ... the truncated factorisation is still what mode 2 runs
        else:
            assert 1 < eng.beam_rank <= max_rank
        got = fit.chi2_batch(P)
        np.testing.assert_allclose(got, ref, rtol=1e-5)
        np.testing.assert_allclose(got[[0, 33]], want, rtol=1e-5)
        assert np.max(np.abs(got / ref - 1)) < 10 * tol          # the singular tail bounds the error
        assert not np.array_equal(got, ref)                        # and it IS the truncated beam
        assert np.array_equal(fit.chi2_batch(P[::-1])[::-1], got)
        fit.close()
    with pytest.raises(Exception, match='beam_tol'):
        synth.make_fit('C2', bpa=20.0, N=64, NV=8, beam_tol=0.5).engine


def test_c3_properties():
    """BASELINE config 3 size (512x512x128).  Properties that hold at any size:
    walkers are independent (order / batch split do not matter, bitwise);
    the truth parameters beat every random walker; chi2 of the truth is about
    the number of independent beams; one walker checked against the oracle."""
    from slam_amd import synth
    fit = synth.make_fit('C3')
    P = synth.walkers(96)
    got = fit.chi2_batch(P)
    perm = np.random.default_rng(0).permutation(len(P))
    assert np.array_equal(fit.chi2_batch(P[perm]), got[perm])
    assert np.array_equal(np.concatenate([fit.chi2_batch(P[:37]), fit.chi2_batch(P[37:])]), got)
    truth = np.array([[synth.TRUTH[k] for k in fit.paramkeys]])
    c_truth = fit.chi2_batch(truth)[0]
    assert c_truth < got.min()
    nbeam = fit.data_valid.size / fit.pixperbeam
    assert 0.8 * nbeam < c_truth < 1.2 * nbeam          # noise-only residual
    s = orc.make_setup(fit.x, fit.y, fit.v, fit.data, bmaj=fit.bmaj, bmin=fit.bmin, bpa=fit.bpa,
                       sigma=fit.sigma, pa=2.0, incl=60.0, rmax=synth.RMAX, signmajor=1,
                       signminor=1)
    np.testing.assert_allclose(got[5], orc.chi2_of(s, P[5]), rtol=1e-9)
    # the cube's truth comes from this package's own cubemodel (synth.make_fit): its
    # chi2 is checked against the oracle too, so that the property is not self-referential
    np.testing.assert_allclose(c_truth, orc.chi2_of(s, truth[0]), rtol=1e-9)
    # dense sky on the same fit: a large broad-lined disk lights two thirds of the
    # conv windows and overflows k_profile's sized launch into k_profile_rest (the launch was
    # sized by the sparse batches above); one walker against the oracle, the batch for order
    D = synth.walkers(40, workload='dense_disk')
    gd = fit.chi2_batch(D)
    assert np.array_equal(fit.chi2_batch(D[::-1])[::-1], gd)
    np.testing.assert_allclose(gd[3], orc.chi2_of(s, D[3]), rtol=1e-9)
    st = fit.engine.chi2_stats(D[3:4])
    assert st['live_tiles'][0] > 0.5 * st['tiles'][0]            # it IS the dense case
    fit.close()


def test_c3_envelope_workload_vs_oracle():
    """`envelope=True`, Ienv = 1 at C3 size: nothing is culled by radius (37 % of the conv
    windows lit); one walker against the oracle, batch order bitwise."""
    from slam_amd import synth
    fit = synth.make_fit('C3', envelope=True)
    E = synth.walkers(24, workload='envelope')
    ge = fit.chi2_batch(E)
    assert np.array_equal(fit.chi2_batch(E[::-1])[::-1], ge)
    s = orc.make_setup(fit.x, fit.y, fit.v, fit.data, bmaj=fit.bmaj, bmin=fit.bmin, bpa=fit.bpa,
                       sigma=fit.sigma, pa=2.0, incl=60.0, rmax=synth.RMAX, signmajor=1,
                       signminor=1, envelope=True)
    np.testing.assert_allclose(ge[7], orc.chi2_of(s, E[7]), rtol=1e-9)
    fit.close()


@pytest.mark.parametrize('beam_px', [(12, 9), (24, 18)])
def test_large_beams_pick_other_conv_tiles(beam_px):
    """K = 39 and K = 77 (BASELINE config 4's beam) select the 64-row/160 KB and
    the 32-row LDS tile variants of the convolution; both against the oracle."""
    from slam_amd import synth
    fit = synth.make_fit('big', N=128, NV=8, beam_px=beam_px)
    assert fit.engine.separable_beam
    P = synth.walkers(5)
    P[:, 0] *= 0.2                     # keep the emission inside the 8 channels
    got = fit.engine.chi2_stats(P)
    s = orc.make_setup(fit.x, fit.y, fit.v, fit.data, bmaj=fit.bmaj, bmin=fit.bmin, bpa=fit.bpa,
                       sigma=fit.sigma, pa=2.0, incl=60.0, rmax=synth.RMAX, signmajor=1,
                       signminor=1)
    for i in range(len(P)):
        st = {}
        model = orc.cubemodel(s, P[i], stages=st)
        np.testing.assert_allclose(got['gf'][i], st['gf'], rtol=1e-9)
        np.testing.assert_allclose(got['ff'][i], st['ff'], rtol=1e-9)
        np.testing.assert_allclose(got['chi2'][i], orc.chi2(s, model), rtol=1e-9)
    assert _golden.relmax(fit.engine.iconv(P[0]), st_conv(s, P[0])) < 1e-9
    fit.close()


def st_conv(s, p):
    st = {}
    orc.cubemodel(s, p, stages=st)
    return st['Iconv']


def test_wide_velocity_gap_uses_exact_index_path():
    """Two velocity windows 184 channels apart inside one 16-channel block:
    beyond the span the fixed-point table index of k_profile is checked for at
    create time, so the kernel's plain double-precision index path runs; it
    must agree with the oracle like the fast one."""
    from slam_amd.channelfit import ChannelFit
    n, nv = 24, 200
    dpix = 12.
    x = -(np.arange(n) - n // 2) * dpix
    y = (np.arange(n) - n // 2) * dpix
    v = (np.arange(nv) - nv / 2 + 0.5) * 0.04
    data = np.random.default_rng(8).normal(0, 1e-3, (nv, n, n))
    vlim = (v[0] - 0.01, v[7] + 0.01, v[-8] - 0.01, v[-1] + 0.01)
    kw = dict(bmaj=4 * dpix, bmin=3 * dpix, bpa=0., sigma=1e-3, pa=10., incl=50., rmax=120.,
              vlim=vlim, signmajor=1, signminor=1)
    fit = ChannelFit(progressbar=False)
    fit.set_cube(x, y, v, data, kw['bmaj'], kw['bmin'], kw['bpa'], kw['sigma'])
    quiet(fit.makegrid, cubefits=None, pa=kw['pa'], incl=kw['incl'], rmax=kw['rmax'],
          vlim=vlim, signmajor=1, signminor=1)
    assert len(fit.v_valid) == 16
    s = orc.make_setup(x, y, v, data, **kw)
    rng = np.random.default_rng(9)
    P = np.zeros((4, 13))
    P[:, 0] = rng.uniform(0.2, 1.0, 4)      # Mstar
    P[:, 1] = rng.uniform(40, 150, 4)       # Rc
    P[:, 2] = [0.4, 0.1, 0.02, 0.3]         # cs
    P[:, 3] = rng.uniform(0, 0.4, 4)
    P[:, 4] = rng.uniform(0, 0.4, 4)
    P[:, 5] = rng.uniform(0, 2, 4)
    ref = np.array([orc.chi2_of(s, p) for p in P])
    np.testing.assert_allclose(fit.chi2_batch(P), ref, rtol=1e-9)
    g = fit.engine.iunif(P[0])
    st = {}
    orc.cubemodel(s, P[0], stages=st)
    assert np.array_equal(g != 0, st['Iunif'] != 0)
    assert _golden.relmax(g, st['Iunif']) < 1e-12
    fit.close()


def test_empty_model_with_nan_data_follows_reference():
    """ff == 0 is tested before anything else (:489): an empty model (every
    sample masked by Rin) has scale 0 whatever gf is, so with a NaN in the cube
    chi2 = nansum(data^2)/sigma^2/pixperbeam, not the 0 of an all-NaN model."""
    from slam_amd import HipLikelihood
    g = _golden.load('F')
    kw = _golden.inputs(g)
    kw['data'] = kw['data'].copy()
    kw['data'][3, 5, 7] = np.nan
    s = orc.make_setup(**kw)
    p_lit = g['vec_deep_p']
    p_empty = p_lit.copy()
    p_empty[6] = 1e6                              # Rin beyond the grid
    expect = np.nansum(kw['data']**2) / s['sigma']**2 / s['pixperbeam']
    np.testing.assert_allclose(orc.chi2_of(s, p_empty), expect, rtol=1e-14)
    d = dict(s)
    d['vunit'] = orc.VUNIT
    with HipLikelihood(d) as eng:
        got = eng.chi2_batch(np.array([p_empty, p_lit, p_empty]))
    np.testing.assert_allclose(got[[0, 2]], expect, rtol=1e-12)
    assert got[1] == 0.0 == orc.chi2_of(s, p_lit)
    # the same without the NaN
    s2 = orc.make_setup(**_golden.inputs(g))
    d2 = dict(s2)
    d2['vunit'] = orc.VUNIT
    with HipLikelihood(d2) as eng:
        np.testing.assert_allclose(eng.chi2_batch(p_empty[None])[0], orc.chi2_of(s2, p_empty), rtol=1e-12)


def test_calls_on_different_streams_are_ordered(c2):
    """Every entry point of a context works in one scratch set: a batch enqueued
    on a side stream and not waited for, followed at once by calls on the
    context's own stream, must give the serial results."""
    import torch
    from slam_amd import synth
    eng = c2.engine
    P = synth.walkers(192, seed=11)
    ref = eng.chi2_batch(P)
    cube = eng.cubemodel(P[3])
    Pd = torch.as_tensor(P, device='cuda')
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        with torch.cuda.stream(side):
            out1 = eng.chi2_batch_torch(Pd)       # side stream, not waited for
        got2 = eng.chi2_batch(P[::-1])            # context stream, at once
        with torch.cuda.stream(side):
            out3 = eng.chi2_batch_torch(Pd)
        cube2 = eng.cubemodel(P[3])               # single-vector product in the same scratch
        side.synchronize()
        assert np.array_equal(out1.cpu().numpy(), ref)
        assert np.array_equal(got2, ref[::-1])
        assert np.array_equal(out3.cpu().numpy(), ref)
        assert np.array_equal(cube2, cube)


def test_folded_regrid_equals_conv_tile_plus_regrid(c2):
    """The fused kernel folds the bilinear regrid into the beam taps for walkers whose
    observed pixels step through consecutive model cells at one fraction (every walker of
    this fit).  Conv mode 3 switches that off: the same walkers through the convolved
    tile and the bilinear regrid (:475-483) must give the same sums."""
    from slam_amd import synth
    eng = c2.engine
    P = synth.walkers(192, seed=23)
    P[:, 8] = np.random.default_rng(23).uniform(-40, 40, len(P))     # xoff, yoff: any fraction,
    P[:, 9] = np.random.default_rng(24).uniform(-40, 40, len(P))     # windows at the image edge
    P[0, 8:10] = 0.0                                                 # fraction exactly 0
    P[1, 8:10] = (c2.x[1] - c2.x[0]) * 3, (c2.y[1] - c2.y[0]) * -2   # whole pixels
    folded = eng.chi2_stats(P)
    eng.set_conv_mode(3)
    try:
        plain = eng.chi2_stats(P)
    finally:
        eng.set_conv_mode(0)
    for key in ('gf', 'ff', 'chi2'):
        np.testing.assert_allclose(folded[key], plain[key], rtol=1e-11, err_msg=key)
    assert np.array_equal(eng.chi2_stats(P)['chi2'], folded['chi2'])  # and back


def test_row_owning_kernel_equals_phase_kernel(c2):
    """k_conv_rows (waves own the window's rows: two barriers per window) is the default for
    one-term beams of 25..32 padded taps in double; conv mode 4 runs k_conv_chi2's fixed-pitch
    instance on the same windows.  Same operations in the same order per output: bitwise equal,
    with the folded regrid and without (mode 3), on sparse and dense sky."""
    from slam_amd import synth
    eng = c2.engine
    P = np.concatenate([synth.walkers(48), synth.walkers(16, workload='dense_disk')])
    P[5, 8:10] = (3.3, -7.1)                    # a shifted walker: observed / model grids out of step
    got = c2.chi2_batch(P)
    st = eng.chi2_stats(P)
    eng.set_conv_mode(4)
    old = c2.chi2_batch(P)
    st4 = eng.chi2_stats(P)
    eng.set_conv_mode(3)
    unf = c2.chi2_batch(P)                      # rows kernel, conv tile + bilinear regrid
    eng.set_conv_mode(0)
    assert np.array_equal(got, old)
    assert np.array_equal(st['gf'], st4['gf']) and np.array_equal(st['ff'], st4['ff'])
    np.testing.assert_allclose(unf, got, rtol=1e-11)
    assert np.all(np.isfinite(got))


def test_row_owning_kernel_80_taps_equals_phase_kernel():
    """The same for BASELINE config 4's beam (K = 77: 80 padded taps, 48-row windows, one
    workgroup per CU): k_conv_rows<48, 80> against k_conv_chi2's 48-row instance (mode 4) and
    against the conv tile + bilinear regrid (mode 3), windows at every image edge, shifted
    walkers (second window list) among the folded ones.  The waves split the output rows
    differently (8 x 6 instead of 6 x 8), so the window sums add in another order: 1e-12."""
    from slam_amd import synth
    fit = synth.make_fit('big', N=192, NV=12, beam_px=(24, 18))
    eng = fit.engine
    ws = eng.window_shape()
    assert (ws['rows'], ws['kpad']) == (48, 80)
    P = np.concatenate([synth.walkers(20), synth.walkers(6, workload='dense_disk')])
    P[:, 0] *= 0.3                              # keep the emission inside the 12 channels
    P[3, 8:10] = (3.3, -7.1)
    P[4, 8:10] = (-55.0, 61.0)
    got = eng.chi2_stats(P)
    eng.set_conv_mode(4)
    old = eng.chi2_stats(P)
    eng.set_conv_mode(3)
    unf = eng.chi2_stats(P)
    eng.set_conv_mode(0)
    assert np.all(np.isfinite(got['chi2'])) and np.all(got['ff'] > 0)
    for key in ('gf', 'ff', 'chi2'):
        np.testing.assert_allclose(got[key], old[key], rtol=1e-12, err_msg=key)
        np.testing.assert_allclose(unf[key], got[key], rtol=1e-11, err_msg=key)
    assert np.array_equal(eng.chi2_stats(P)['chi2'], got['chi2'])
    fit.close()


def test_profile_list_longer_than_expected(c2):
    """k_profile's one-item-per-workgroup launch is sized from the list lengths of recent
    launch groups; a list that turns out longer is finished by k_profile_rest.  Sparse
    batches first (short lists), then a dense one on the same context: bitwise the result
    of a context that sized its launch for everything."""
    from slam_amd import synth
    sparse = synth.walkers(64, seed=3)
    dense = synth.walkers(64, seed=4, workload='dense_disk')
    ref_dense, ref_sparse = c2.chi2_batch(dense), c2.chi2_batch(sparse)
    fit = synth.make_fit('C2')
    try:
        first = fit.chi2_batch(sparse)                    # no count back yet: full launch
        again = fit.chi2_batch(sparse)                    # sized from the first call's list
        got = fit.chi2_batch(dense)                       # list several times longer than the hint
        assert np.array_equal(first, ref_sparse) and np.array_equal(again, ref_sparse)
        assert np.array_equal(got, ref_dense)
        assert np.array_equal(fit.chi2_batch(dense), ref_dense)
    finally:
        fit.close()


@pytest.mark.timeout(1100)
def test_c4_full_size():
    """BASELINE config 4 at full size: 1024x1024x256 cube, K = 77 beam (32-row
    conv windows), 4096 walkers.  Batch order and split are bitwise immaterial,
    the truth beats every walker, and one walker is checked against the oracle
    (channel-chunked to bound its memory: ~2 minutes of one host core)."""
    from slam_amd import synth
    fit = synth.make_fit('C4')
    eng = fit.engine
    assert eng.nv == 256 and eng.nx == eng.ny == 1024 and fit.gaussbeam.shape == (77, 77)
    assert eng.separable_beam and eng.beam_rank == 1
    assert 1 <= eng.walkers_per_group <= 128
    P = synth.walkers(4096)
    got = fit.chi2_batch(P)
    assert np.all(np.isfinite(got)) and np.all(got > 0)
    sub = np.random.default_rng(1).permutation(4096)[:300]
    assert np.array_equal(fit.chi2_batch(P[sub]), got[sub])
    assert np.array_equal(np.concatenate([fit.chi2_batch(P[:11]), fit.chi2_batch(P[11:64])]), got[:64])
    truth = np.array([[synth.TRUTH[k] for k in fit.paramkeys]])
    c_truth = fit.chi2_batch(truth)[0]
    assert c_truth < got.min()
    nbeam = fit.data_valid.size / fit.pixperbeam
    assert 0.8 * nbeam < c_truth < 1.2 * nbeam
    s = orc.make_setup(fit.x, fit.y, fit.v, fit.data, bmaj=fit.bmaj, bmin=fit.bmin, bpa=fit.bpa,
                       sigma=fit.sigma, pa=2.0, incl=60.0, rmax=synth.RMAX, signmajor=1,
                       signminor=1)
    i = 7
    st = {}
    model = orc.cubemodel(s, P[i], stages=st, kchunk=16)
    np.testing.assert_allclose(got[i], orc.chi2(s, model), rtol=1e-9)
    gst = eng.chi2_stats(P[i][None])
    np.testing.assert_allclose(gst['gf'][0], st['gf'], rtol=1e-9)
    np.testing.assert_allclose(gst['ff'][0], st['ff'], rtol=1e-9)
    fit.close()
