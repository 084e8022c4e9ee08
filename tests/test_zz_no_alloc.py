"""slamhip_chi2_batch_static (the allocation-free entry point) against
slamhip_chi2_batch_dev, through the C ABI: two calls in a row, the second with
a smaller input, then a smaller fit whose scratch overlays the first one's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fit(n, nv, beam_px=(5, 3.5)):
    from slam_amd import synth
    return synth.make_fit('static', N=n, NV=nv, beam_px=beam_px)


def test_static_entry_matches_allocating_entry():
    import torch
    from slam_amd import synth
    fit = _fit(96, 24)
    eng = fit.engine
    total, need1, wg = eng.static_info()
    assert total == 64 << 20 and 0 < need1 <= total and wg >= 1
    P = torch.as_tensor(synth.walkers(3 * wg + 5), device='cuda')      # several launch groups
    ref = eng.chi2_batch_torch(P)
    out = torch.full((P.shape[0],), -1.0, dtype=torch.float64, device='cuda')
    eng.chi2_batch_static_torch(P, out=out)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    # second call, smaller input: the array still holds the first call's contents
    P2 = torch.as_tensor(synth.walkers(7, seed=5), device='cuda')
    out2 = torch.full((7,), -1.0, dtype=torch.float64, device='cuda')
    eng.chi2_batch_static_torch(P2, out=out2)
    ref2 = eng.chi2_batch_torch(P2)
    torch.cuda.synchronize()
    assert torch.equal(out2, ref2)
    # a smaller fit on the same GPU shares the array (another layout over stale bytes)
    small = _fit(48, 12)
    P3 = torch.as_tensor(synth.walkers(9, seed=6), device='cuda')
    out3 = small.engine.chi2_batch_static_torch(P3)
    ref3 = small.engine.chi2_batch_torch(P3)
    torch.cuda.synchronize()
    assert torch.equal(out3, ref3)
    # and the first fit again afterwards
    out4 = eng.chi2_batch_static_torch(P2)
    torch.cuda.synchronize()
    assert torch.equal(out4, ref2)
    # conv mode 3 (conv tile + bilinear regrid for every walker, no folded taps) runs from the
    # static scratch too; the dense mode 1 is refused
    from slam_amd._lib import SlamhipError
    eng.set_conv_mode(3)
    ref5 = eng.chi2_batch_torch(P2)
    out5 = eng.chi2_batch_static_torch(P2)
    torch.cuda.synchronize()
    assert torch.equal(out5, ref5)
    np.testing.assert_allclose(ref5.cpu().numpy(), ref2.cpu().numpy(), rtol=1e-11)
    eng.set_conv_mode(1)
    with pytest.raises(SlamhipError, match='conv modes 0, 2 and 3'):
        eng.chi2_batch_static_torch(P2)
    eng.set_conv_mode(0)
    small.close()
    fit.close()


def test_static_entry_rejects_oversize_fit():
    import torch
    from slam_amd import synth
    from slam_amd._lib import SlamhipError
    fit = _fit(384, 96, beam_px=(8, 6))
    total, need1, wg = fit.engine.static_info()
    assert need1 > total and wg == 0
    P = torch.as_tensor(synth.walkers(2), device='cuda')
    with pytest.raises(SlamhipError, match='static scratch holds'):
        fit.engine.chi2_batch_static_torch(P)
    # the allocating entry point is unaffected
    assert np.all(np.isfinite(fit.engine.chi2_batch_torch(P).cpu().numpy()))
    fit.close()
