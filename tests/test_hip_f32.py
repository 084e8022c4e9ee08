"""GPU parity of the single-precision modes (slamhip_desc::precision 1 and 2)
against the golden vectors recorded from the reference and against the double
mode.  north_star's tolerance for fp32 is 1e-3 relative: chi2, and cubes by
max|d|/max|ref|.

precision 1 ('f32'): profile accumulation, Iunif and beam passes in float;
geometry and table index in double, so every sample selects the reference's
table entry and the cubes agree everywhere.
precision 2 ('f32g'): geometry in float too; a few samples in 1e5 land in the
neighbouring table entry, so its cubes are checked in rms and by the share of
voxels off by more than 1e-3 of the peak; chi2 to 1e-3 like precision 1."""
import numpy as np
import pytest

import _golden
from oracle import channelfit_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-3


def engine(s, precision):
    from slam_amd import HipLikelihood
    d = dict(s)
    d['vunit'] = orc.VUNIT
    return HipLikelihood(d, precision=precision)


@pytest.mark.parametrize('config', _golden.CONFIGS)
@pytest.mark.parametrize('precision', ['f32', 'f32g'])
def test_chi2_and_cubes_vs_reference(config, precision):
    g = _golden.load(config)
    s = orc.make_setup(**_golden.inputs(g))
    names = _golden.vectors(g)
    P = np.array([g[f'vec_{n}_p'] for n in names])
    ref = np.array([float(g[f'vec_{n}_chi2']) for n in names])
    with engine(s, precision) as eng:
        got = eng.chi2_batch(P)
        np.testing.assert_allclose(got, ref, rtol=TOL)
        assert np.array_equal(eng.chi2_batch(P[::-1])[::-1], got)       # order-independent, bitwise
        for n in [m for m in names if f'vec_{m}_Iunif' in g.files][:4]:
            iu, refu = eng.iunif(g[f'vec_{n}_p']), g[f'vec_{n}_Iunif']
            peak = np.max(np.abs(refu))
            if peak == 0:
                assert not iu.any()
                continue
            if precision == 'f32':
                assert np.array_equal(iu != 0, refu != 0)
                assert _golden.relmax(iu, refu) < 1e-5
                assert _golden.relmax(eng.cubemodel(g[f'vec_{n}_p']), g[f'vec_{n}_model']) < TOL
            else:
                err = np.abs(iu - refu) / peak
                assert np.sqrt(np.mean(err**2)) < TOL
                assert np.mean(err > TOL) < 1e-3


def test_c2_f32_matches_f64():
    """BASELINE config 2 cube, 64 walkers: both float modes against double."""
    from slam_amd import synth
    P = synth.walkers(64)
    fit = synth.make_fit('C2')
    ref = fit.chi2_batch(P)
    iu64 = fit.engine.iunif(P[3])
    data = fit.data
    fit.close()
    for precision, tol_cube in (('f32', 1e-5), ('f32g', None)):
        f = synth.make_fit('C2', precision=precision)
        assert np.array_equal(f.data, data)
        got = f.chi2_batch(P)
        np.testing.assert_allclose(got, ref, rtol=TOL)
        # the row-owning kernel (default) and the phase kernel (conv mode 4) do the same float
        # operations in the same order per output
        f.engine.set_conv_mode(4)
        assert np.array_equal(f.chi2_batch(P), got)
        f.engine.set_conv_mode(0)
        iu = f.engine.iunif(P[3])
        if tol_cube:
            assert _golden.relmax(iu, iu64) < tol_cube
        else:
            err = np.abs(iu - iu64) / np.max(iu64)
            assert np.sqrt(np.mean(err**2)) < 1e-4 and np.mean(err > TOL) < 1e-4
        f.close()


@pytest.mark.parametrize('beam_px', [(12, 9), (24, 18)])
def test_f32_large_beams(beam_px):
    """K = 39 and K = 77: the float tile is half the size, so other window
    variants are picked than in double; chi2 against the double mode."""
    from slam_amd import synth
    P = synth.walkers(5)
    P[:, 0] *= 0.2
    f64 = synth.make_fit('big', N=128, NV=8, beam_px=beam_px)
    ref = f64.chi2_batch(P)
    f64.close()
    f32 = synth.make_fit('big', N=128, NV=8, beam_px=beam_px, precision='f32')
    np.testing.assert_allclose(f32.chi2_batch(P), ref, rtol=TOL)
    f32.close()


def test_f32_rotated_beam_and_general_regrid():
    """Direct taps (rotated beam) and the unfolded regrid (dx != dy golden
    configuration is covered above; here an offset walker set) in float."""
    from slam_amd import synth
    P = synth.walkers(6)
    P[:, 8] = [0.0, 1.3, -2.1, 0.4, 7.7, -0.2]     # xoff
    P[:, 9] = [0.0, -0.6, 3.3, 0.1, -4.2, 0.9]     # yoff
    for bpa in (0.0, 20.0):
        f64 = synth.make_fit('rot', N=96, NV=24, beam_px=(5, 3.5), bpa=bpa)
        ref = f64.chi2_batch(P)
        f64.close()
        f32 = synth.make_fit('rot', N=96, NV=24, beam_px=(5, 3.5), bpa=bpa, precision='f32')
        np.testing.assert_allclose(f32.chi2_batch(P), ref, rtol=TOL)
        f32.close()


def test_f32_rejects_what_it_does_not_cover():
    from slam_amd import synth
    from slam_amd._lib import SlamhipError
    with pytest.raises(SlamhipError, match='uniform'):
        synth.make_fit('x', N=48, NV=12, beam_px=(5, 3.5), scaling='mom0ft', precision='f32').engine
    f = synth.make_fit('x', N=48, NV=12, beam_px=(5, 3.5), precision='f32')
    with pytest.raises(SlamhipError, match='conv mode 1'):
        f.engine.set_conv_mode(1)
    f.close()
