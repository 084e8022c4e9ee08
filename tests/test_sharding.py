"""Walker sharding over a process group: world_size 2 and 3 on CPU (gloo).
The local likelihood is a deterministic stand-in; what is checked is the
scatter/gather plumbing: block partition with remainder, rank-0 authority,
bit-identical results for any world size."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from slam_amd.sharding import ShardedLikelihood, partition


def fake_chi2(P):
    P = np.asarray(P)
    return np.sum(P * np.arange(1, P.shape[1] + 1), axis=1) + np.prod(P[:, :3], axis=1)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, nw, out_dir, nparam=13):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        sh = ShardedLikelihood(fake_chi2, device='cpu', nparam=nparam)
        P = np.random.default_rng(5).random((nw, nparam)) if rank == 0 else None
        got = sh.chi2(P, nw).numpy()
        np.save(os.path.join(out_dir, f'r{rank}.npy'), got)
    finally:
        dist.destroy_process_group()


def test_partition():
    assert partition(7, 2) == ([4, 3], [0, 4])
    assert partition(2, 4) == ([1, 1, 0, 0], [0, 1, 2, 2])
    c, o = partition(1024, 8)
    assert c == [128] * 8 and o[-1] == 896


@pytest.mark.parametrize('world,nw', [(2, 7), (3, 10), (2, 1)])
def test_sharded_equals_single(tmp_path, world, nw):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, nw, str(tmp_path)), nprocs=world, join=True)
    ref = fake_chi2(np.random.default_rng(5).random((nw, 13)))
    for r in range(world):
        got = np.load(tmp_path / f'r{r}.npy')
        assert np.array_equal(got, ref), r       # bitwise, on every rank


def test_sharded_six_parameter_rows(tmp_path):
    """The mock-PV likelihood's 6-parameter rows take the same route."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 9, str(tmp_path), 6), nprocs=2, join=True)
    ref = fake_chi2(np.random.default_rng(5).random((9, 6)))
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f'r{r}.npy'), ref), r


def test_single_process_passthrough():
    sh = ShardedLikelihood(fake_chi2, device='cpu')
    P = np.random.default_rng(1).random((5, 13))
    assert np.array_equal(sh.chi2(P).numpy(), fake_chi2(P))


# ---------------------------------------------------------------------------
# The real engine under two ranks (both on cuda:0, gloo between them): what a
# node with fewer GPUs than ranks can rehearse of the N > 1 path.
def _golden_fit():
    import contextlib
    import io

    import _golden
    from slam_amd.channelfit import ChannelFit
    g = _golden.load('A')
    kw = _golden.inputs(g)
    fit = ChannelFit(disk=kw['disk'], envelope=kw['envelope'], progressbar=False)
    fit.set_cube(kw['x'], kw['y'], kw['v'], kw['data'], kw['bmaj'], kw['bmin'], kw['bpa'], kw['sigma'])
    with contextlib.redirect_stdout(io.StringIO()):
        fit.makegrid(cubefits=None, pa=kw['pa'], incl=kw['incl'], rmax=kw['rmax'], vlim=kw['vlim'],
                     nlayer=kw['nlayer'], gaussmargin=kw['gaussmargin'],
                     signmajor=kw['signmajor'], signminor=kw['signminor'])
    P = np.array([g[f'vec_{n}_p'] for n in _golden.vectors(g)])
    return fit, np.concatenate([P, P[::-1], P[:3]])


def _gpu_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        fit, P = _golden_fit()
        sh = ShardedLikelihood(fit.engine.chi2_batch_torch, device='cuda')
        assert sh.host_wire
        got = sh.chi2(P if rank == 0 else None, len(P))
        assert got.is_cuda
        np.save(os.path.join(out_dir, f'g{rank}.npy'), got.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_real_engine_two_ranks_equal_one(tmp_path):
    fit, P = _golden_fit()
    ref = fit.chi2_batch(P)
    port = _free_port()
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f'g{r}.npy'), ref), r     # bitwise, on every rank


def _nccl_worker(rank, world, port, out_dir):
    """One rank over RCCL (backend "nccl"): the device-side collectives of the N > 1 path,
    as far as one GPU can run them."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    try:
        fit, P = _golden_fit()
        sh = ShardedLikelihood(fit.engine.chi2_batch_torch, device='cuda')
        assert not sh.host_wire and sh.world == 1
        # the collectives themselves, on device tensors, as chi2() issues them for world > 1
        t = torch.as_tensor(P, device='cuda')
        dist.broadcast(t, src=0)
        loc = fit.engine.chi2_batch_torch(t)
        out = torch.empty_like(loc)
        dist.all_gather_into_tensor(out, loc)
        dist.barrier()
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, 'n0.npy'), out.cpu().numpy())
        np.save(os.path.join(out_dir, 'n0s.npy'), sh.chi2(P, len(P)).cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_rccl_collectives_on_one_rank(tmp_path):
    """backend 'nccl' (= RCCL) with the calls the sharded path makes: broadcast of the
    parameters, all-gather of chi2, barrier -- on a one-rank group, which is all a one-GPU
    box can hold (two ranks on one device are refused by RCCL)."""
    fit, P = _golden_fit()
    ref = fit.chi2_batch(P)
    mp.spawn(_nccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert np.array_equal(np.load(tmp_path / 'n0.npy'), ref)
    assert np.array_equal(np.load(tmp_path / 'n0s.npy'), ref)
