"""`mom0*` scalings of ChannelFit: host-side deconvolved
moment-0 maps (CPU, bit-equal to the reference) and the reordered forward model
/ chi2 on the GPU against golden vectors recorded from the reference
(tests/golden/make_golden_mom0.py)."""
import contextlib
import io
import os

import numpy as np
import pytest

import _golden

CASES = [('A', 'mom0ft'), ('A', 'mom0clean'), ('A', 'mom0model'),
         ('B', 'mom0ft'), ('B', 'mom0clean')]


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def gold(cfg):
    return (_golden.load(cfg),
            np.load(os.path.join(_golden.GOLDEN_DIR, f'channelfit_mom0_{cfg}.npz')))


_fits = {}


def product_fit(cfg, scaling):
    if (cfg, scaling) in _fits:
        return _fits[cfg, scaling]
    from slam_amd.channelfit import ChannelFit
    g, _ = gold(cfg)
    kw = _golden.inputs(g)
    fit = ChannelFit(disk=kw['disk'], envelope=kw['envelope'], scaling=scaling,
                     progressbar=False)
    fit.set_cube(kw['x'], kw['y'], kw['v'], kw['data'], kw['bmaj'], kw['bmin'],
                 kw['bpa'], kw['sigma'])
    quiet(fit.makegrid, cubefits=None, pa=kw['pa'], incl=kw['incl'], rmax=kw['rmax'],
          vlim=kw['vlim'], nlayer=kw['nlayer'], gaussmargin=kw['gaussmargin'],
          signmajor=kw['signmajor'], signminor=kw['signminor'])
    _fits[cfg, scaling] = fit
    return fit


@pytest.mark.parametrize('cfg,sc', CASES)
def test_host_deconvolved_moment0_bitwise(cfg, sc):
    """makegrid's mom0 section == the reference's, bit for bit."""
    _, m = gold(cfg)
    fit = product_fit(cfg, sc)
    assert np.array_equal(fit.gaussbeam, m[f'{sc}_gaussbeam'])
    assert np.array_equal(fit.mom0, m[f'{sc}_mom0'])
    assert fit.sigma_mom0 == m[f'{sc}_sigma_mom0']
    assert np.array_equal(fit.mom0decon, m[f'{sc}_mom0decon'])
    assert np.array_equal(fit.resdecon, m[f'{sc}_resdecon'])
    d = fit.setup_dict()
    assert d['mom0decon'] is fit.mom0decon and d['gaussbeam'] is fit.gaussbeam


def test_unknown_scaling_is_refused():
    from slam_amd.channelfit import ChannelFit
    with pytest.raises(ValueError):
        ChannelFit(scaling='mom0')


def test_clean_loads_and_saves_components(tmp_path):
    from slam_amd import deconv
    rng = np.random.default_rng(0)
    t = np.arange(9) - 4
    beam = np.exp2(-4 * ((t[:, None] / 3.)**2 + (t[None, :] / 3.)**2))
    img = np.zeros((20, 22))
    img[8, 9] = 5.
    from scipy.signal import convolve
    img = convolve(img, beam, mode='same') + rng.normal(0, 0.01, img.shape)
    path = str(tmp_path / 'cc.txt')
    cc = deconv.clean(img, beam, sigma=0.01, savetxt=path, verbose=False)
    # flux is conserved: components + residual/beamarea
    assert np.sum(cc) * np.sum(beam) == pytest.approx(np.sum(img), rel=0.2)
    assert np.unravel_index(np.argmax(cc), cc.shape) == (8, 9)
    assert np.allclose(quiet(deconv.clean, img, beam, 0.01, loadtxt=path), cc)


# ---------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('cfg,sc', CASES)
def test_gpu_model_and_chi2_match_reference(cfg, sc):
    _, m = gold(cfg)
    fit = product_fit(cfg, sc)
    names = [str(n) for n in m[f'{sc}_vector_names']]
    P = np.array([m[f'{sc}_vec_{n}_p'] for n in names])
    ref = np.array([float(m[f'{sc}_vec_{n}_chi2']) for n in names])
    chi2 = fit.chi2_batch(P)
    np.testing.assert_allclose(chi2, ref, rtol=1e-9)
    # batch split / order independence, bitwise
    assert np.array_equal(fit.chi2_batch(P[::-1])[::-1], chi2)
    assert np.array_equal(np.array([fit.chi2_batch(p[None])[0] for p in P]), chi2)
    # the direct K x K taps (conv_mode 1) and the factorised beam agree
    fit.engine.set_conv_mode(1)
    np.testing.assert_allclose(fit.chi2_batch(P), chi2, rtol=1e-11)
    np.testing.assert_allclose(fit.chi2_batch(P), ref, rtol=1e-9)
    fit.engine.set_conv_mode(0)
    for n in names:
        pre = f'{sc}_vec_{n}_'
        p = dict(zip(fit.paramkeys, m[pre + 'p']))
        model, st = fit.engine.cubemodel(m[pre + 'p'], return_stats=True)
        assert st['chi2'] == pytest.approx(float(m[pre + 'chi2']), rel=1e-9)
        scale = np.max(np.abs(m[pre + 'model_chsum']))
        assert np.max(np.abs(model.sum(axis=(1, 2)) - m[pre + 'model_chsum'])) < 1e-9 * scale
        if pre + 'model' in m:
            assert _golden.relmax(fit.cubemodel(**p), m[pre + 'model']) < 1e-9
        if pre + 'noconv_model' in m:
            nc = fit.cubemodel(**p, convolving=False)
            assert _golden.relmax(nc, m[pre + 'noconv_model']) < 1e-9


@pytest.mark.gpu
def test_gpu_mom0_fitting_runs(tmp_path):
    """The fit driver works unchanged on a mom0 scaling: lnprob = -chi2/2."""
    _, m = gold('A')
    fit = product_fit('A', 'mom0ft')
    fixed = {'cs': 0.4, 'Rin': 0, 'Ienv': 0, 'xoff': 0, 'yoff': 0, 'voff': 0,
             'incloff': 0, 'paoff': 0}
    quiet(fit.fitting, Mstar_range=[0.1, 1.0], Rc_range=[30, 300], pI_range=[0, 3],
          h1_range=[0, 0.5], h2_range=[0, 0.5], fixed_params=fixed,
          filename=str(tmp_path / 'f'),
          kwargs_emcee_corner={'nwalkers_per_ndim': 2, 'nburnin': 3, 'nsteps': 3, 'seed': 1})
    p = m['mom0ft_vec_generic_p']
    lnp = fit.lnprob(np.array([np.log10(p[0]), np.log10(p[1]), p[3], p[4], p[5]]))
    assert lnp == pytest.approx(-0.5 * float(m['mom0ft_vec_generic_chi2']), rel=1e-9)
    assert os.path.exists(tmp_path / 'f.popt.txt')


@pytest.mark.gpu
def test_gpu_sparse_path_equals_dense_on_a_multi_segment_image():
    """200 x 200 pixels = 4 row segments of 64 (the last one partial), several beam
    windows per plane: the batch path (lit channel ranges, row segments stored only
    where they hold emission, listed windows taken by persistent workgroups) against
    the dense direct-tap path (conv mode 1: every pixel stored, every window run),
    with offsets that push part of the image outside the model grid, and with what
    earlier batches left in the scratch cube."""
    from slam_amd import synth
    fit = synth.make_fit(None, N=200, NV=12, beam_px=(6, 4), scaling='mom0ft')
    try:
        P = synth.walkers(7)
        P[1, 8], P[1, 9] = 35.0, -20.0          # xoff, yoff [au]
        P[2, 8], P[2, 9] = -150.0, 120.0
        P[3, 10] = 0.3                          # voff
        P[4, 1] = 400.0                         # disk larger than the image
        chi2 = fit.chi2_batch(P)
        assert np.all(np.isfinite(chi2)) and len(set(chi2.tolist())) == len(P)
        # other walkers' cubes are in the scratch slots now: order and split, bitwise
        assert np.array_equal(fit.chi2_batch(P[::-1])[::-1], chi2)
        assert np.array_equal(np.array([fit.chi2_batch(p[None])[0] for p in P]), chi2)
        assert np.array_equal(fit.chi2_batch(P[2:5]), chi2[2:5])
        fit.engine.set_conv_mode(1)
        dense = fit.chi2_batch(P)
        fit.engine.set_conv_mode(0)
        np.testing.assert_allclose(chi2, dense, rtol=1e-11)
        # single-vector product (one workgroup per window, writes the cube) agrees too
        for i in (0, 2):
            _, st = fit.engine.cubemodel(P[i], return_stats=True)
            assert st['chi2'] == pytest.approx(chi2[i], rel=1e-12)
    finally:
        fit.close()


@pytest.mark.gpu
@pytest.mark.parametrize('npix', [256, 200])
def test_gpu_row_owning_beam_kernel_on_the_observed_grid(npix):
    """One-term beam of 25..32 padded taps on a square observed grid: the batch path lists the
    64 x 64 windows with emission (k_compact) and runs the row-owning beam kernel on them
    (k_conv_rows<OBS>: only the stored row segments of a staged tile are used).  Against the
    older tile kernel (conv mode 4), the dense direct taps (mode 1) and the single-vector
    product; offsets push emission across window and image edges."""
    from slam_amd import synth
    # 200 = three whole windows and one of 8 pixels per axis
    fit = synth.make_fit(None, N=npix, NV=16, beam_px=(8, 6), scaling='mom0ft')
    try:
        eng = fit.engine
        assert eng.window_shape()['rows'] == 64 and eng.window_shape()['kpad'] == 32
        P = np.concatenate([synth.walkers(9), synth.walkers(3, workload='dense_disk')])
        P[:, 0] *= 0.3                          # keep the emission inside the 16 channels
        P[1, 8], P[1, 9] = 35.0, -20.0
        P[2, 8], P[2, 9] = -150.0, 120.0
        P[3, 10] = 0.3
        P[4, 1] = 400.0
        chi2 = fit.chi2_batch(P)
        counts = eng.group_counts(len(P))
        assert counts['live_windows'] > 0 and counts['window_tiles'] > 0 and counts['window_rows'] > 0   # the list carries tile bitmaps and row masks
        assert np.all(np.isfinite(chi2)) and len(set(chi2.tolist())) == len(P)
        assert np.array_equal(fit.chi2_batch(P[::-1])[::-1], chi2)
        assert np.array_equal(np.array([fit.chi2_batch(p[None])[0] for p in P]), chi2)
        eng.set_conv_mode(4)
        tile = fit.chi2_batch(P)
        eng.set_conv_mode(1)
        dense = fit.chi2_batch(P)
        eng.set_conv_mode(0)
        np.testing.assert_allclose(chi2, tile, rtol=1e-11)
        np.testing.assert_allclose(chi2, dense, rtol=1e-11)
        for i in (0, 2, 10):
            _, st = eng.cubemodel(P[i], return_stats=True)
            assert st['chi2'] == pytest.approx(chi2[i], rel=1e-11)
    finally:
        fit.close()


@pytest.mark.gpu
def test_gpu_row_owning_beam_kernel_on_a_rectangular_grid():
    """The same on a 256 x 176 pixel cube (rows cut from a square synthetic one): the observed
    grid's rows and columns are told apart in the window list, the staging and the sums."""
    from slam_amd import synth
    from slam_amd.channelfit import ChannelFit
    sq = synth.make_fit(None, N=256, NV=16, beam_px=(8, 6), scaling='mom0ft')
    x, y, v, data = sq.x.copy(), sq.y[40:216].copy(), sq.v.copy(), sq.data[:, 40:216, :].copy()
    bmaj, bmin, bpa, sigma = sq.bmaj, sq.bmin, sq.bpa, sq.sigma
    sq.close()
    fit = ChannelFit(scaling='mom0ft', progressbar=False)
    fit.set_cube(x, y, v, data, bmaj, bmin, bpa, sigma)
    quiet(fit.makegrid, cubefits=None, pa=2.0, incl=60.0, rmax=synth.RMAX, signmajor=1, signminor=1)
    try:
        eng = fit.engine
        assert (eng.nx, eng.ny) == (256, 176) and eng.window_shape()['rows'] == 64
        P = synth.walkers(8)
        P[:, 0] *= 0.3
        P[1, 8], P[1, 9] = 35.0, -60.0
        P[2, 1] = 400.0
        chi2 = fit.chi2_batch(P)
        assert eng.group_counts(len(P))['window_rows'] > 0
        assert np.all(np.isfinite(chi2)) and len(set(chi2.tolist())) == len(P)
        eng.set_conv_mode(4)
        tile = fit.chi2_batch(P)
        eng.set_conv_mode(1)
        dense = fit.chi2_batch(P)
        eng.set_conv_mode(0)
        np.testing.assert_allclose(chi2, tile, rtol=1e-11)
        np.testing.assert_allclose(chi2, dense, rtol=1e-11)
        assert np.array_equal(fit.chi2_batch(P[::-1])[::-1], chi2)
    finally:
        fit.close()


# ---------------------------------------------------------------------------
# Moment maps behind plotmom: golden vectors recorded from the
# reference's makemom01 by tests/golden/make_golden_moments.py.
def moments_gold():
    return np.load(os.path.join(_golden.GOLDEN_DIR, 'channelfit_moments.npz'))


@pytest.mark.parametrize('cfg', ['A', 'B'])
def test_host_moment_maps_bitwise(cfg):
    """makegrid's observed mom0 / mom1 and makemom01 on the reference's own
    model cubes == the reference's maps, bit for bit (NaN pattern included)."""
    from slam_amd.channelfit import makemom01
    mg = moments_gold()
    g = _golden.load(cfg)
    fit = product_fit(cfg, 'uniform')
    assert np.array_equal(fit.mom0, mg[f'{cfg}_obs_mom0'])
    assert np.array_equal(fit.mom1, mg[f'{cfg}_obs_mom1'], equal_nan=True)
    assert fit.sigma_mom0 == mg[f'{cfg}_sigma_mom0']
    for n in (str(s) for s in mg[f'{cfg}_vector_names']):
        if f'vec_{n}_model' not in g.files:
            continue
        m = makemom01(g[f'vec_{n}_model'], fit.v_valid, fit.sigma)
        assert np.array_equal(m['mom0'], mg[f'{cfg}_vec_{n}_mom0'])
        assert np.array_equal(m['mom1'], mg[f'{cfg}_vec_{n}_mom1'], equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', ['A', 'B'])
def test_gpu_moment_maps(cfg):
    """slamhip_moments: bit-equal to numpy's reduction of the cube the GPU itself
    returns (same order of additions), and equal to the reference's maps to the
    cubes' own 1e-9 (pixels within 1e-9 of the 3 sigma_mom0 blanking threshold may
    fall on either side of it)."""
    from slam_amd.channelfit import PARAMKEYS as KEYS, makemom01
    mg = moments_gold()
    fit = product_fit(cfg, 'uniform')
    clip = 3 * fit.sigma_mom0
    for n in (str(s) for s in mg[f'{cfg}_vector_names']):
        p = dict(zip(KEYS, mg[f'{cfg}_vec_{n}_p']))
        mom0, mom1 = fit.modelmom('mod', **p)
        own = makemom01(fit.cubemodel(**p), fit.v_valid, fit.sigma)
        assert np.array_equal(mom0, own['mom0'])
        assert np.array_equal(mom1, own['mom1'], equal_nan=True)
        ref0, ref1 = mg[f'{cfg}_vec_{n}_mom0'], mg[f'{cfg}_vec_{n}_mom1']
        assert _golden.relmax(mom0, ref0) < 1e-9
        edge = np.abs(ref0 - clip) < 1e-9 * clip
        assert np.array_equal(np.isnan(mom1) | edge, np.isnan(ref1) | edge)
        both = np.isfinite(mom1) & np.isfinite(ref1)
        assert both.sum() > 100
        np.testing.assert_allclose(mom1[both], ref1[both], rtol=0, atol=1e-9)
        # 'res' = observed - model, 'obs' = the maps makegrid kept
        r0, r1 = fit.modelmom('res')
        assert np.array_equal(r0, fit.mom0 - mom0)
        assert np.array_equal(r1, fit.mom1 - mom1, equal_nan=True)
        assert fit.modelmom('obs')[0] is fit.mom0
    with pytest.raises(ValueError):
        fit.modelmom('nothing')
