#!/usr/bin/env python3
"""Benchmark: batched channelfit log-likelihood (model cube + chi-square) on
SYN-v1 synthetic cubes.  One "step" = one batched likelihood
call as the sampler sees it: `--walkers` parameter vectors per GPU in, their
chi2 out (weak scaling: per-GPU work is fixed as --gpus grows).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12      # B/s, MI355X spec (MI355X_MICROARCH.md, chip-level parameters)
# FP64 vector peak: half of the guide's 157.3 TFLOP/s FP32 vector peak (a wave64 FP64 FMA issues in
# 4 cycles on a SIMD-32, an FP32 one in 2): 256 CUs x 4 SIMDs x 16 FMA/clk x 2 flop x 2.4 GHz
FP64_PEAK = 78.6e12
FP32_PEAK = 157.3e12   # guide's chip table, vector FP32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default='C3', choices=['C2', 'C3', 'C4', 'P5'],
                    help='C2-C4: channelfit SYN-v1 cubes; P5: pvfitting mock-PV likelihood on the '
                         'reference test PV diagrams (BASELINE config 5)')
    ap.add_argument('--walkers', type=int, default=1024, help='walkers per GPU per step')
    ap.add_argument('--bpa', type=float, default=0.0, help='beam position angle [deg]; 0 = separable')
    ap.add_argument('--scaling', default='uniform', choices=['uniform', 'mom0ft', 'mom0clean'],
                    help="ChannelFit scaling (C2-C4); the headline metric is 'uniform'")
    ap.add_argument('--workload', default='sparse', choices=['sparse', 'dense_disk', 'envelope'],
                    help="walker set (slam_amd/synth.py): 'sparse' = the headline (example priors); "
                         "'dense_disk' / 'envelope' light most of the sky")
    ap.add_argument('--dtype', default='f64', choices=['f64', 'f32', 'f32g'],
                    help="arithmetic of the hot path: f64 = the reference's (the headline); f32 = float "
                         "accumulation / beam passes with double geometry and index; f32g = float geometry too")
    ap.add_argument('--beam-tol', type=float, default=0.0,
                    help='opt-in truncated factorisation of a rotated beam (slamhip_desc::beam_tol); '
                         '0 = exact (the default and the headline)')
    ap.add_argument('--conv-mode', type=int, default=0, choices=[0, 2, 3, 4],
                    help='diagnostic: 2 = factorised beam even where direct taps are the default, '
                         '3 = no folded regrid, 4 = k_conv_chi2 where k_conv_rows is the default')
    ap.add_argument('--cpu-evals', type=int, default=1,
                    help='oracle evaluations timed for cpu_baseline (0 = skip)')
    ap.add_argument('--cpu-procs', type=int, default=-1,
                    help='processes of the multi-core CPU baseline (-1 = auto: OMP_NUM_THREADS '
                         'capped at 16; 0 = skip); runs before the GPU is touched')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='process-group backend; gloo only to rehearse N>1 on a box with fewer GPUs')
    ap.add_argument('--cpu-worker', type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker >= 0:        # one process of the multi-core CPU baseline
        print(_cpu_worker((args.config, args.bpa, args.scaling, args.cpu_worker, args.cpu_evals)))
        return

    if os.environ.get('SLAM_BENCH_WATCHDOG'):     # a run that hangs says where, then ends
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['SLAM_BENCH_WATCHDOG']), exit=True)

    multi = None
    if (int(os.environ.get('WORLD_SIZE', 1)) == 1 and args.cpu_evals > 0 and args.cpu_procs != 0
            and args.config != 'P5'):
        multi = cpu_baseline_multi(args)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    if args.backend == 'gloo':          # rehearsal: ranks may share a GPU
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group('gloo')

    from slam_amd import synth
    from slam_amd.sharding import ShardedLikelihood
    if args.config == 'P5':
        run_pv(args, rank, world, torch, dist)
        if world > 1:
            dist.destroy_process_group()
        return
    t_start = time.perf_counter()
    fit = synth.make_fit(args.config, bpa=args.bpa, scaling=args.scaling,
                         envelope=args.workload == 'envelope', precision=args.dtype,
                         beam_tol=args.beam_tol)
    eng = fit.engine
    if args.conv_mode:
        eng.set_conv_mode(args.conv_mode)
    torch.cuda.synchronize()
    startup_s = time.perf_counter() - t_start   # per rank: synthetic cube, makegrid, context
    nw_total = args.walkers * world
    P = synth.walkers(nw_total, workload=args.workload) if rank == 0 else None
    sharded = ShardedLikelihood(eng.chi2_batch_torch, device='cuda')
    Pd = torch.as_tensor(P, device='cuda') if rank == 0 else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        chi2 = sharded.chi2(Pd, nw_total)
    barrier()
    t0 = time.perf_counter()                    # the headline: nothing but the K steps in here
    for _ in range(args.steps):
        chi2 = sharded.chi2(Pd, nw_total)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt, startup_s], dtype=torch.float64,
                         device='cpu' if args.backend == 'gloo' else 'cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, startup_s = float(t[0].item()), float(t[1].item())
    chi2 = chi2.cpu().numpy()

    if rank == 0:
        # per-kernel durations: a pass of its own (HIP events around every launch,
        # on the launch stream), outside the timed region
        mine = Pd[:args.walkers].contiguous()
        psteps = max(1, min(args.steps, 3))
        eng.profile_begin()
        for _ in range(psteps):
            eng.chi2_batch_torch(mine)
        torch.cuda.synchronize()
        kern = eng.profile_end()
        # one launch group on its own, then what it touched and computed
        wl = min(eng.walkers_per_group, args.walkers)
        eng.chi2_batch_torch(mine[:wl].contiguous())
        torch.cuda.synchronize()
        try:
            counts = eng.group_counts(wl)
            counts['walkers'] = wl
        except Exception as e:                  # dense conv mode etc.: no accounting
            print('bench: no group counts:', e, file=sys.stderr)
            counts = None
        out = report(args, fit, eng, world, nw_total, dt, chi2, kern, psteps, counts)
        out['config']['startup_s_per_rank'] = round(startup_s, 3)   # max over ranks, outside the timed region
        if world == 1 and args.cpu_evals > 0:
            out['cpu_baseline'] = cpu_baseline(fit, P, chi2, args.cpu_evals)
            if multi is not None:
                out['cpu_baseline']['all_cores'] = multi
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def kernel_model(eng, fit, counts, itemsize=8, geomsize=8, scaling='uniform'):
    """Per launch group: the bytes each hot kernel MUST move in this design and the
    USEFUL floating-point operations it needs, from slamhip_group_counts and
    slamhip_window_shape.  Useful = real taps only: K per axis, K+1 when
    the regrid is folded into the beam; the kernels run the tap count padded to a multiple of 8
    (`kpad`), and that padding is not counted.  Tiles are 32x8 pixels; a geometry entry is
    {vlos, weight} = 2 values."""
    B = itemsize
    W = counts['walkers']
    K = fit.gaussbeam.shape[0]
    ws = eng.window_shape()
    kpad, ty, tx = ws['kpad'], ws['rows'], ws['cols']
    rank = eng.beam_rank
    nslot = eng.nleaf                          # >= samples; padding ignored
    tile = 32 * 8 * B
    geom = counts['geom_groups'] * 64 * 2 * geomsize
    by = {
        # reads each sample's coordinates once per 8 walkers, writes the stored entries + flag words
        'k_surface': -(-W // 8) * nslot * (4 + 2 * 8) + geom + W * (nslot // 64) * 4,
        # reads the stored entries (once: the channel blocks of a tile share an L2), writes lit tiles + flags
        'k_profile': geom + counts['lit_planes'] * (tile + 1),
    }
    fl = {
        # table lookups: 16 channels x (index add + multiply-add) per hit
        'k_profile': counts['profile_hits'] * 16 * 3,
    }
    if scaling == 'uniform':
        # reads every lit tile once, its descriptor, and the data under a live window once per
        # block of 8 walkers (k_compact's list order: the block's windows are neighbours in the
        # list and the XCD's L2 serves the others; a lower bound, since not every walker of a
        # block has the window live)
        by['k_conv'] = counts['lit_planes'] * tile + counts['live_windows'] * (
            (tx - 1) * (ty - 1) * 8 / min(8, W) + 240)
        if rank >= 1:
            taps = K + 1 if (ws['fold'] and rank == 1) else K
            # row pass: every row with emission, tx outputs x taps; column pass: executed blocks of
            # 8 output rows x 8 tap rows x 64 lanes, of which taps / kpad are real taps
            fl['k_conv'] = 2 * rank * (counts['window_rows'] * tx * taps
                                       + counts['col_blocks'] * 64 * 8 * 8 * taps / kpad)
        else:
            # direct taps: executed (output-row block, 8-tap-row block) pairs, once per tap column,
            # rb x 8 multiply-adds per lane each; K of kpad tap rows are real
            rb = 16 if ty == 64 else 8
            fl['k_conv'] = 2 * 64 * rb * 8 * K * counts['col_blocks'] * K / kpad
    else:
        # 'mom0*': the regridded, rescaled cube lives on the observed grid.  Its lit 32x8 tiles are
        # not counted one by one; taken as the model's lit tiles scaled by the grids' pixel ratio
        # (the same sky, regridded).
        obs_tiles = counts['lit_planes'] * (eng.nx * eng.ny) / (eng.nneed * eng.nneed)
        # k_mom_model + k_regrid_mom0: the lit Iunif tiles twice (moment, then the scaled
        # output), the moment map written and read, the stored row segments written
        by['k_regrid'] = 2 * counts['lit_planes'] * tile + W * eng.nneed**2 * 16 + obs_tiles * tile
        fl['k_regrid'] = counts['lit_planes'] * 256 * 1 + obs_tiles * 256 * 10    # moment sum; bilinear + scale per stored value
        if counts['window_rows'] > 0:
            # row-owning beam kernel on the listed 64x64 windows (k_conv_rows<OBS>): as the
            # 'uniform' path's model with the plain K taps
            fl['k_conv'] = 2 * (counts['window_rows'] * tx * K + counts['col_blocks'] * 64 * 8 * 8 * K / kpad)
            by['k_conv'] = obs_tiles * tile + counts['live_windows'] * (64 * 64 * 8 + 240)
        else:
            # older tile kernel (conv mode 4, beams it still serves): dense inside every listed
            # window (row pass over the window + halo rows, column pass over its rows)
            terms = max(rank, 1)
            fl['k_conv'] = 2 * terms * counts['live_windows'] * tx * K * ((ty + K - 1) + ty)
            by['k_conv'] = counts['live_windows'] * ((ty + K - 1) * (tx + K - 1) + ty * tx) * 8
    return by, fl


def report(args, fit, eng, world, nw_total, dt, chi2, kern, psteps, counts):
    B = 8 if args.dtype == 'f64' else 4
    nv, n, nx, ny = eng.nv, eng.nneed, eng.nx, eng.ny
    bytes_eval = synth_bytes_per_eval(eng, B)
    # SURVEY.md 8d's DENSE algorithmic bytes per walker of each kernel (four stages)
    dense = {'k_profile': B * nv * n * n,
             'k_conv': 2 * B * nv * n * n,
             'k_regrid': B * nv * (n * n + nx * ny)}
    if args.scaling != 'uniform':
        dense['k_regrid'] = B * nv * (n * n + 3 * nx * ny)
        dense['k_conv'] = 2 * B * nv * nx * ny
    elif kern.get('k_regrid', (0, 0))[1] == 0:
        dense['k_conv'] += dense.pop('k_regrid')       # fused: conv + regrid + chi2 in one kernel
    evals_prof = args.walkers * psteps
    total_ms = sum(v[0] for v in kern.values())
    model_b, model_f = (kernel_model(eng, fit, counts, B, 4 if args.dtype == 'f32g' else 8, args.scaling)
                        if counts else ({}, {}))
    peak_flops = FP64_PEAK if args.dtype == 'f64' else FP32_PEAK
    valu = 'fp64_valu' if args.dtype == 'f64' else 'fp32_valu'
    groups = -(-args.walkers // eng.walkers_per_group)          # launch groups per step
    per_kernel = {}
    for name, (ms, cnt) in kern.items():
        if cnt == 0:
            continue
        d = {'ms_per_launch': ms / cnt, 'launches': cnt, 'ms_per_eval': ms / evals_prof,
             'share': ms / total_ms}
        sec = ms * 1e-3 / cnt
        if name in dense:
            d['algorithmic_x_peak'] = dense[name] * evals_prof / (ms * 1e-3) / HBM_PEAK
        if counts and counts['walkers'] == min(eng.walkers_per_group, args.walkers) and groups * psteps == cnt:
            # the counted group is a typical launch (same walkers per launch)
            if name in model_b:
                d['bytes_per_launch'] = model_b[name]
                d['hbm_GBps'] = model_b[name] / sec / 1e9
                d['hbm_frac'] = model_b[name] / sec / HBM_PEAK
            if name in model_f:
                d['flops_per_launch'] = model_f[name]         # useful operations (float ones at f32)
                d['TFLOPs'] = model_f[name] / sec / 1e12
                d['valu_frac'] = model_f[name] / sec / peak_flops
        per_kernel[name] = d
    dom = max((k for k in per_kernel if k in dense), key=lambda k: kern[k][0])
    dk = per_kernel[dom]
    sec = dk['ms_per_launch'] * 1e-3
    traffic, traffic_note = measured_traffic(dom, args, eng)
    roofline = {'kernel': dom, 'ms_per_launch': dk['ms_per_launch'], 'launches': dk['launches'],
                'traffic': traffic, 'traffic_source': traffic_note,
                'algorithmic_x_peak': dk.get('algorithmic_x_peak'),
                'per_kernel': per_kernel,
                'pipeline': {'dense_bytes_per_eval': bytes_eval,
                             'algorithmic_x_peak': bytes_eval * nw_total * args.steps / world / dt / HBM_PEAK,
                             'kernel_ms_per_eval': total_ms / evals_prof}}
    if traffic is not None:
        roofline['traffic_frac'] = traffic / sec / HBM_PEAK
    if 'flops_per_launch' in dk and 'bytes_per_launch' in dk:
        # Both roofs of the dominant kernel; the BOUND is the one its arithmetic intensity
        # selects: useful operations per byte it moves (the measured HBM-side traffic of the
        # committed PMC passes when they describe this build and workload, else the bytes the
        # design must move) against the machine balance peak_flops / HBM_PEAK.
        F, Bc = dk['flops_per_launch'], dk['bytes_per_launch']
        Bm = traffic if traffic is not None else Bc
        ai, balance = F / Bm, peak_flops / HBM_PEAK
        roofs = {valu: {'achieved': F / sec / 1e12, 'peak': peak_flops / 1e12, 'unit': 'TFLOP/s',
                        'frac': F / sec / peak_flops, 'useful_flops_per_launch': F},
                 'hbm': {'achieved': Bm / sec / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                         'frac': Bm / sec / HBM_PEAK, 'bytes_per_launch': Bm,
                         'bytes': 'measured' if traffic is not None else 'compulsory (model)',
                         'compulsory_bytes_per_launch': Bc, 'compulsory_frac': Bc / sec / HBM_PEAK}}
        bound = 'hbm' if ai < balance else valu
        roofline.update({'bound': bound, 'achieved': roofs[bound]['achieved'], 'peak': roofs[bound]['peak'],
                         'unit': roofs[bound]['unit'], 'frac': roofs[bound]['frac'],
                         'arithmetic_intensity_flop_per_byte': ai, 'machine_balance_flop_per_byte': balance,
                         'roofs': roofs, 'counts': counts})
    else:
        roofline.update({'bound': 'hbm', 'achieved': None, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                         'frac': None})
    return {
        'metric': 'walker chi2 evals/sec (whole node) at 512x512x128'
                  if args.config == 'C3' and args.scaling == 'uniform' and args.workload == 'sparse' and args.dtype == 'f64'
                  else f'walker chi2 evals/sec (whole node), {args.config} scaling={args.scaling} '
                       f'workload={args.workload} dtype={args.dtype}',
        'value': nw_total * args.steps / dt, 'unit': 'evals/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64' if args.dtype == 'f64' else 'f32',
        'data': 'synthetic',
        'config': {'workload': f'SYN-v1 {args.config} ({args.workload}, {args.dtype}): {nx}x{ny}x{nv} cube, '
                               f'{args.walkers} walkers per GPU per step, beam K={fit.gaussbeam.shape[0]} '
                               f'bpa={args.bpa:g} (beam rank {eng.beam_rank})'
                               + (f' beam_tol={args.beam_tol:g}' if args.beam_tol else '')
                               + (f' conv_mode={args.conv_mode}' if args.conv_mode else ''),
                   'walkers_per_gpu': args.walkers, 'walkers_per_launch': eng.walkers_per_group,
                   'npix': eng.npix, 'nneed': n, 'nleaf': eng.nleaf,
                   'chi2_checksum': float(np.sum(chi2)),
                   'build': build_id()},
        'roofline': roofline,
    }


def build_id():
    """Source hash + stamps state of the libslamhip.so this process loaded
    (slamhip_version(); equals slam_amd.build.src_sha('slamhip') of the tree it was built from)."""
    from slam_amd import _lib
    w = _lib.build_words(_lib.load().slamhip_version().decode())
    return f"{w.get('src')} stamps={w.get('stamps')} variant={w.get('variant')}"


def synth_bytes_per_eval(e, itemsize=8):
    """write Iunif; read Iunif + write Iconv; read Iconv + read data."""
    return itemsize * e.nv * (4 * e.nneed**2 + e.nx * e.ny)


def run_pv(args, rank, world, torch, dist):
    """pvfitting's mock-PV likelihood: the example's grid
    (59 x 21 x 287 parent, six 2x nestings) and the reference's test PV diagrams
    (tests/golden/pvfit_example_fit.npz holds them as data), `--walkers` parameter
    vectors per GPU per step in the example's priors."""
    from slam_amd import synth
    from slam_amd.sharding import ShardedLikelihood
    mpvd, kw = synth.make_pvfit()
    nw_total = args.walkers * world
    Q = synth.pv_walkers(nw_total)
    eng = mpvd.engine(Q[0, 1], kw['incl'], kw['linewidth'], kw['rin'], kw['rout'])
    sharded = ShardedLikelihood(eng.lnprob_batch_torch, device='cuda', nparam=6)
    Qd = torch.as_tensor(Q, device='cuda') if rank == 0 else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        lnp = sharded.chi2(Qd, nw_total)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lnp = sharded.chi2(Qd, nw_total)
    barrier()
    dt = time.perf_counter() - t0
    eng.profile_begin()                         # per-kernel events: a pass of their own
    for _ in range(args.steps):
        lnp = sharded.chi2(Qd, nw_total)
    barrier()
    kern = eng.profile_end()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cpu' if args.backend == 'gloo' else 'cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    lnp = lnp.cpu().numpy()
    if rank != 0:
        return
    evals_rank = args.walkers * args.steps
    nv = eng.nv
    ncol = sum(nx * ny for nx, ny, _ in eng.shapes)
    # algorithmic bytes per evaluation of k_pv_tau: every live cell's constants
    # (int32 + 4 doubles) and its 5 table entries once, plus one column-sum
    # image per level and axis
    alg_tau = eng.live_cells * (36 + 40) + 2 * nv * ncol * 8
    total_ms = sum(v[0] for v in kern.values())
    per_kernel = {n: {'ms_per_launch': ms / c, 'launches': c, 'ms_per_eval': ms / evals_rank,
                      'share': ms / total_ms} for n, (ms, c) in kern.items() if c}
    ms, cnt = kern['k_pv_tau']
    achieved = alg_tau * evals_rank / (ms * 1e-3)
    traffic = pv_traffic(eng)
    out = {
        'metric': 'walker lnprob evals/sec (whole node), pvfitting mock-PV on test.pvfitting.*.fits',
        'value': nw_total * args.steps / dt, 'unit': 'evals/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
        'data': 'reference test PV diagrams (as golden data) + synthetic walkers',
        'config': {'workload': f'P5: mock-PV likelihood, parent grid {eng.shapes[0]}, '
                               f'{len(eng.shapes)} levels, {eng.live_cells} live cells (both axes), '
                               f'nv={nv}, {args.walkers} walkers per GPU per step',
                   'walkers_per_gpu': args.walkers, 'walkers_per_launch': eng.walkers_per_group,
                   'lnp_checksum': float(np.sum(lnp))},
        # k_pv_tau is latency bound: its cell constants and tables are
        # served from L2 / MALL, so the algorithmic bytes are not HBM bytes; `frac` is the
        # measured HBM traffic of the committed PMC passes against the peak
        'roofline': {'bound': 'hbm', 'kernel': 'k_pv_tau',
                     'achieved': None if traffic is None else traffic / (ms / cnt * 1e-3) / 1e9,
                     'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                     'frac': None if traffic is None else traffic / (ms / cnt * 1e-3) / HBM_PEAK,
                     'algorithmic_x_peak': achieved / HBM_PEAK,
                     'traffic': traffic, 'bytes_per_launch': alg_tau * evals_rank / cnt,
                     'ms_per_launch': ms / cnt, 'launches': cnt, 'per_kernel': per_kernel},
    }
    if world == 1 and args.cpu_evals > 0:
        out['cpu_baseline'] = cpu_baseline_pv(kw, Q, lnp, max(args.cpu_evals, 8))
    print(json.dumps(out), flush=True)


def pv_traffic(eng):
    """HBM bytes per k_pv_tau launch from the committed rocprofv3 PMC passes of
    this same command (profiles/r1_pv_traffic.json; FETCH_SIZE x2 + WRITE_SIZE, as
    for the channelfit kernels), or null."""
    try:
        t = json.load(open(os.path.join(ROOT, 'profiles', 'r1_pv_traffic.json')))
    except Exception:
        return None
    if t.get('_meta', {}).get('walkers_per_launch') != eng.walkers_per_group:
        return None
    for k, v in t.items():
        if k.startswith('k_pv_tau'):
            return 2.0 * v['fetch_bytes_per_launch_max'] + v['write_bytes_per_launch_max']
    return None


def cpu_baseline_pv(kw, Q, lnp_gpu, nevals):
    """The numpy oracle of the reference's mock-PV path, one process."""
    from oracle import pvfitting_oracle as orc
    from slam_amd import synth
    s, obs = synth.make_pvfit_oracle(orc, rc_first=Q[0, 1])
    orc.lnprob(s, obs, Q[0])                         # tables / first-call work untimed
    t0 = time.perf_counter()
    ref = np.array([orc.lnprob(s, obs, Q[i]) for i in range(nevals)])
    dt = time.perf_counter() - t0
    err = float(np.max(np.abs(ref - lnp_gpu[:nevals]) / np.abs(ref)))
    return {'value': nevals / dt, 'unit': 'evals/s', 'cores': 1, 'kind': 'port',
            'sample': f'first {nevals} of the {len(Q)} walkers, oracle/pvfitting_oracle.py '
                      '(numpy+scipy, single process)',
            'seconds': dt, 'max_rel_err_vs_gpu': err}


def _cpu_worker(job):
    """One process of the multi-core CPU baseline: its own oracle setup on a
    noise cube of the workload's shape (the cost does not depend on the data
    values), then `nevals` likelihood evaluations."""
    config, bpa, scaling, seed, nevals = job
    from oracle import channelfit_oracle as orc
    from slam_amd import synth
    n, nv, bmaj_px, bmin_px = synth.CONFIGS[config]
    x, y, v, dpix = synth.axes(n, nv)
    data = np.random.default_rng(seed).normal(0.0, synth.SIGMA, (nv, n, n))
    s = orc.make_setup(x, y, v, data, bmaj=bmaj_px * dpix, bmin=bmin_px * dpix, bpa=bpa,
                       sigma=synth.SIGMA, pa=2.0, incl=60.0, rmax=synth.RMAX, signmajor=1,
                       signminor=1, scaling=scaling)
    P = synth.walkers(1024)[seed * nevals:(seed + 1) * nevals]
    t0 = time.perf_counter()
    for p in P:
        orc.chi2_of(s, p)
    return time.perf_counter() - t0


def cpu_baseline_multi(args):
    """The reference's only parallel mode is one walker per process
    (utils.py:50-55): P independent oracle processes, one evaluation each per
    `--cpu-evals`, timed as a whole.  Runs before this process initialises the
    GPU; the workers are separate `python bench.py --cpu-worker i` processes."""
    import subprocess
    procs = args.cpu_procs
    if procs < 0:
        procs = min(16, int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1)))
    if procs < 2:
        return None
    env = dict(os.environ, OPENBLAS_NUM_THREADS='1', OMP_NUM_THREADS='1', MKL_NUM_THREADS='1')
    cmd = [sys.executable, os.path.abspath(__file__), '--config', args.config, '--bpa',
           str(args.bpa), '--scaling', args.scaling, '--cpu-evals', str(args.cpu_evals)]
    t0 = time.perf_counter()
    kids = [subprocess.Popen(cmd + ['--cpu-worker', str(i)], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, env=env, text=True) for i in range(procs)]
    inner = []
    try:
        for k in kids:
            out, _ = k.communicate(timeout=600)
            if k.returncode != 0:
                return None
            inner.append(float(out.strip().splitlines()[-1]))
    except (subprocess.TimeoutExpired, ValueError, IndexError):
        for k in kids:
            if k.poll() is None:
                k.kill()
        return None
    wall = time.perf_counter() - t0
    return {'value': procs * args.cpu_evals / max(inner), 'unit': 'evals/s', 'cores': procs,
            'kind': 'port',
            'sample': f'{procs} processes x {args.cpu_evals} evaluation(s), each with its own '
                      'oracle setup on a noise cube of the same shape; rate = evaluations / '
                      'slowest process (setup excluded)',
            'seconds': max(inner), 'wall_seconds_with_setup': wall}


TRAFFIC_PROFILE = 'r3_traffic.json'


def measured_traffic(kernel, args, eng):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled as
    MI355X_MICROARCH.md's HBM section prescribes for gfx950; calibration on
    kernels of known size is in DESIGN.md; Infinity-Cache hits are included in
    FETCH_SIZE, so this is an upper bound on HBM bytes).  bench.py itself cannot
    read the counters, so this is the figure of the profiled run of the same
    command BY THE SAME BUILD: the profile carries the source hash of the library
    it was collected with (`_meta.src_sha`), and a profile of other sources, another
    workload or another launch-group size is refused (null + the reason)."""
    path = os.path.join(ROOT, 'profiles', TRAFFIC_PROFILE)
    try:
        t = json.load(open(path))
    except Exception:
        return None, f'no profiles/{TRAFFIC_PROFILE}'
    meta = t.get('_meta', {})
    from slam_amd import build as _build
    if meta.get('src_sha') != _build.src_sha('slamhip'):
        return None, (f"stale profile: profiles/{TRAFFIC_PROFILE} was collected with src "
                      f"{meta.get('src_sha')}, this build is {_build.src_sha('slamhip')}")
    want = {'config': args.config, 'walkers_per_launch': eng.walkers_per_group, 'bpa': args.bpa,
            'scaling': args.scaling, 'workload': args.workload, 'dtype': args.dtype}
    got = {k: meta.get(k) for k in want}
    if got != want or args.walkers % eng.walkers_per_group:     # full launch groups only
        return None, f'profile is for {got}'
    # the bench's kernel ids cover what one group launches under that id: the beam pass is
    # k_conv_rows' two launches (folded / unfolded list) or one k_conv_chi2 instance
    names = {'k_conv': ('k_conv_rows', 'k_conv_chi2'), 'k_profile': ('k_profile<', 'k_profile_rest'),
             'k_surface': ('k_surface',), 'k_regrid': ('k_regrid',)}
    per_group = meta.get('launch_groups') or max((v.get('launches', 0) for k, v in t.items()
                                                  if k.startswith(('k_surface',))), default=0)
    total, found = 0.0, False
    for k, v in t.items():
        if k != '_meta' and k.startswith(names.get(kernel, (kernel,))) and v.get('launches', 0) >= max(3, per_group // 2):   # (not a one-off launch of the single-vector entry points)
            total += 2.0 * v['fetch_bytes_per_launch_max'] + v['write_bytes_per_launch_max']
            found = True
    if found:
        return (total,
                f'profiles/{TRAFFIC_PROFILE} (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, separate passes, '
                'max over launches; same src_sha)')
    return None, 'kernel not in profile'


def cpu_baseline(fit, P, chi2_gpu, nevals):
    """Time the CPU oracle (numpy/scipy restatement of the reference path) on
    the first `nevals` walkers of the same workload, one process, one core."""
    from oracle import channelfit_oracle as orc
    from slam_amd import synth
    s = orc.make_setup(fit.x, fit.y, fit.v, fit.data, bmaj=fit.bmaj, bmin=fit.bmin,
                       bpa=fit.bpa, sigma=fit.sigma, pa=2.0, incl=60.0, rmax=synth.RMAX,
                       signmajor=1, signminor=1, scaling=fit.scaling)
    t0 = time.perf_counter()
    ref = np.array([orc.chi2_of(s, P[i]) for i in range(nevals)])
    dt = time.perf_counter() - t0
    err = float(np.max(np.abs(ref - chi2_gpu[:nevals]) / np.abs(ref)))
    return {'value': nevals / dt, 'unit': 'evals/s', 'cores': 1, 'kind': 'port',
            'sample': f'first {nevals} of the {len(P)} walkers, full cube, '
                      'oracle/channelfit_oracle.py (numpy+scipy, single process)',
            'seconds': dt, 'max_rel_err_vs_gpu': err}


if __name__ == '__main__':
    main()
