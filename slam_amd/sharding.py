"""Walker sharding across GPUs: one process per GPU (`torch.distributed`;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

Walkers are independent (each chi2 depends only on its own parameters -- 13 for channelfit, 6 for
the mock-PV likelihood -- and on
read-only per-fit data replicated on every GPU), so the data path needs no
collective: the batch is handed out in contiguous blocks and the chi-square
values are gathered back.  The result is bit-identical for any number of GPUs.
"""
from __future__ import annotations

import numpy as np


def partition(nw: int, world: int):
    """Contiguous block sizes and offsets, remainder spread over low ranks."""
    counts = [nw // world + (1 if r < nw % world else 0) for r in range(world)]
    offs = [0]
    for c in counts[:-1]:
        offs.append(offs[-1] + c)
    return counts, offs


class ShardedLikelihood:
    """chi2 for a global walker batch, sharded over the process group.

    `local_chi2(P_local)` evaluates this rank's block.  With `device='cuda'` it
    receives/returns float64 CUDA tensors (e.g. `HipLikelihood.chi2_batch_torch`),
    otherwise numpy arrays.  Every rank calls `chi2` collectively; rank 0's
    `P` is authoritative (the sampler lives there) and every rank gets the full
    chi2 vector back.
    """

    def __init__(self, local_chi2, device='cuda', group=None, nparam=13):
        self.nparam = nparam
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.local_chi2 = local_chi2
        self.group = group
        self.on_gpu = device == 'cuda'
        self.device = torch.device('cuda', torch.cuda.current_device()) if self.on_gpu \
            else torch.device('cpu')
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
            # gloo is a host backend: its collectives get host tensors even when the
            # likelihood runs on the GPU (all_gather_into_tensor on device tensors hangs there)
            self.host_wire = self.on_gpu and dist.get_backend(group) == 'gloo'
        else:
            self.rank, self.world = 0, 1
            self.host_wire = False
        self.wire = torch.device('cpu') if self.host_wire else self.device

    def chi2(self, P, nw=None):
        torch, dist = self.torch, self.dist
        if nw is None:
            nw = len(P)
        if self.rank == 0:
            Pt = torch.as_tensor(np.ascontiguousarray(P, dtype=np.float64)
                                 if not torch.is_tensor(P) else P,
                                 dtype=torch.float64).to(self.wire)
        else:
            Pt = torch.empty((nw, self.nparam), dtype=torch.float64, device=self.wire)
        if self.world > 1:
            dist.broadcast(Pt, src=0, group=self.group)          # <= 13*8 B per walker
        Pt = Pt.to(self.device)
        counts, offs = partition(nw, self.world)
        cnt, off = counts[self.rank], offs[self.rank]
        mine = Pt[off:off + cnt]
        if self.on_gpu:
            loc = self.local_chi2(mine)
        else:
            loc = torch.as_tensor(np.asarray(self.local_chi2(mine.numpy()), dtype=np.float64))
        if self.world == 1:
            return loc
        maxc = max(counts)
        pad = torch.zeros(maxc, dtype=torch.float64, device=self.wire)
        pad[:cnt] = loc
        out = torch.empty(maxc * self.world, dtype=torch.float64, device=self.wire)
        dist.all_gather_into_tensor(out, pad, group=self.group)   # 8 B per walker
        out = torch.cat([out[r * maxc:r * maxc + counts[r]] for r in range(self.world)])
        return out.to(self.device)
