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
