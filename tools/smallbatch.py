#!/usr/bin/env python3
"""Run ON THE GPU BOX: what ONE GPU of an 8-GPU node is handed when the sampler's
ensemble is small.  The example fit has 80 walkers = 40 per emcee
half-step = 5 per GPU on 8 GPUs.  Measures, at C3 (512x512x128, f64):
  * wall time per call of the device entry point (enqueue + synchronize) and of the
    host-pointer entry point (staging copies included) for 1..128 walkers per call;
  * the two collectives of a sharded step (broadcast of the parameter batch, all-gather
    of chi2) on a ONE-rank RCCL group: their software floor on this box (no xGMI hop).
One JSON object on stdout."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from slam_amd import synth
    torch.cuda.set_device(0)
    fit = synth.make_fit('C3')
    eng = fit.engine
    P = synth.walkers(1024)
    Pd = torch.as_tensor(P, device='cuda')
    out = {'config': 'C3 f64 sparse', 'per_call': {}}
    for nw in (1, 5, 10, 20, 40, 80, 128):
        reps = 60
        rows = [Pd[(i * nw) % 896:(i * nw) % 896 + nw].contiguous() for i in range(reps)]
        for r in rows[:5]:
            eng.chi2_batch_torch(r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in rows:                       # a sampler waits for every half-step's result
            eng.chi2_batch_torch(r)
            torch.cuda.synchronize()
        dev = (time.perf_counter() - t0) / reps
        hrows = [P[(i * nw) % 896:(i * nw) % 896 + nw] for i in range(reps)]
        t0 = time.perf_counter()
        for r in hrows:
            eng.chi2_batch(r)
        host = (time.perf_counter() - t0) / reps
        out['per_call'][nw] = {'dev_entry_us': dev * 1e6, 'host_entry_us': host * 1e6,
                               'dev_us_per_eval': dev * 1e6 / nw, 'evals_per_s': nw / dev}
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29571')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    coll = {}
    for nw in (40, 1024, 4096):
        p = torch.zeros((nw, 13), dtype=torch.float64, device='cuda')
        c = torch.zeros(nw, dtype=torch.float64, device='cuda')
        o = torch.empty(nw, dtype=torch.float64, device='cuda')
        for _ in range(20):
            dist.broadcast(p, src=0)
            dist.all_gather_into_tensor(o, c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            dist.broadcast(p, src=0)
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t0) / 200
        t0 = time.perf_counter()
        for _ in range(200):
            dist.all_gather_into_tensor(o, c)
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / 200
        coll[nw] = {'broadcast_us': tb * 1e6, 'all_gather_us': tg * 1e6}
    out['collectives_one_rank_rccl'] = coll
    dist.destroy_process_group()
    fit.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
