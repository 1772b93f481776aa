#!/usr/bin/env python
"""EFE rollouts/sec on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic input: calculate_G_repeated over 128
rows (32 root frames x 4 actions) with 10 MC samples and depth 5 (BASELINE configs[1]), followed by
the action posterior; at N > 1 every rank runs its own 128 rows (episodes shard with no data-path
collective, weak scaling) and one all_gather of the [32,4] action posteriors per step is the only RCCL
traffic.  Inputs are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAC_ROLLOUT = 6_739_934_560          # 51 encoder + 100 transition + 150 decoder passes
MAC_DECB_ROW = 18_874_368 + 1_179_648   # k_dec_b: ConvTranspose2d(64,32,3,s2) 32*32*9*64*32 + ConvTranspose2d(32,1,3,s1) 64*64*9*32
PEAK_FP32_MFMA_TF = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, fp32 in / fp32 acc
CLASS_MACS_PER_ROW = {               # algorithmic MACs per network row, by kernel class
    'dec_dense_16384': 256 * 16384, 'dec_a_convT1_convT2': 2 * 256 * 9 * 64 * 64, 'dec_b_convT3_final_reduce': MAC_DECB_ROW,
}


def synth_frames(n, device, seed=0):
    """dSprites-like frames: one filled square + the reward bar of game_environment.py:44-54,70-71."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 4, generator=g)
    fr = torch.zeros(n, 1, 64, 64)
    for i in range(n):
        side = 6 + int(u[i, 0] * 18); y = 3 + int(u[i, 1] * (61 - side)); x = int(u[i, 2] * (64 - side))
        fr[i, 0, y:y + side, x:x + side] = 1.0
        r = 2 * float(u[i, 3]) - 1
        if r > 0:
            fr[i, 0, 0:3, 0:32] = r
        else:
            fr[i, 0, 0:3, 32:64] = -r
    return fr.to(device)


def usable_cores():
    """cores this process may actually use: affinity mask, capped by the cgroup CPU quota (os.cpu_count() ignores both,
    and oversubscribing torch's intra-op pool on a quota-limited container stalls it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(depth, samples, target_s=12.0):
    """The oracle (CPU restatement of the reference's own torch op sequence, torch RNG like the reference)
    timed on this host's cores on a bounded sample of the same workload."""
    from oracle import synth
    from oracle.efe_oracle import OracleModel, TorchNoise
    cores = usable_cores()
    torch.set_num_threads(cores)
    m = OracleModel(synth.make_weights(1234, 1.15), TorchNoise())

    def run(rows, d, s):
        o = torch.from_numpy(np.repeat(synth.make_frames(5, (rows + 3) // 4), 4, axis=0)[:rows])
        pi = torch.eye(4).repeat((rows + 3) // 4, 1)[:rows]
        t = time.perf_counter()
        with torch.no_grad():
            m.calculate_G_repeated(o, pi, d, False, s, 0)
        return time.perf_counter() - t
    run(4, 1, 1)                                   # warm-up (thread pools, oneDNN primitives)
    t_unit = run(8, 1, 1)                          # 8 rows x 1 stage x 1 sample
    per_row_full = t_unit / 8 * depth * samples    # estimated seconds per full rollout row
    rows = int(min(128, max(4, target_s / max(per_row_full, 1e-4))))
    rows -= rows % 4
    print(f'[bench] cpu_baseline: {cores} threads, unit pass {t_unit:.3f}s, timing {rows} rows', file=sys.stderr, flush=True)
    dt = run(rows, depth, samples)
    return {'value': rows / dt, 'unit': 'rollouts/s', 'cores': cores, 'kind': 'port',
            'sample': f'{rows} rows x depth {depth} x {samples} MC samples, 1 timed pass after warm-up, torch-CPU eager '
                      f'oracle (oracle/efe_oracle.py, torch RNG like the reference), {dt:.2f} s'}


def bench_mcts(a, model, device, world, rank, dist):
    """BASELINE configs[2]/[3]: E episodes per GPU, each a full MCTS decision (50 expansions with S MC samples,
    simulation depth 5, use_means=False, early stop disabled), planned in lock-step; the root visit distributions
    are gathered across ranks.  One decision ~ 51 expansions x 4 rows x 2.694 GFLOP + 50 x 1.35 GFLOP ~ 617 GFLOP."""
    import daimc_amd
    E = a.episodes
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.use_means, p.threshold, p.samples = 50, 5, False, 2.0, a.samples
    frames = synth_frames(E, device, seed=200 + rank)

    def step():
        out, distn = daimc_amd.active_inference_mcts_batch(model, frames, p, o_shape=(1, 64, 64), episode_offset=rank * E)
        if world > 1:
            daimc_amd.gather_action_posteriors(distn.to(device), world * E)
        return out
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        dec = world * E * a.steps / dt
        gflop_dec = (51 * 4 * a.samples * 134_721_312 * 2 + 50 * (5 * (18_176 + 541_696) + 5 * 134_179_616) * 2) / 1e9
        print(json.dumps({'metric': 'MCTS decisions/sec (50 expansions, %d MC samples, sim depth 5)' % a.samples, 'value': dec,
                          'unit': 'decisions/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                          'config': {'workload': f'lock-step MCTS, {E} episodes per GPU (BASELINE configs[2])', 'episodes_per_gpu': E},
                          'rollout_equivalents_per_s': dec * gflop_dec / 13.480, 'achieved_tflops_total': dec * gflop_dec / 1e3}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--rows', type=int, default=128)
    ap.add_argument('--samples', type=int, default=10)
    ap.add_argument('--depth', type=int, default=5)
    ap.add_argument('--dec-chunk', type=int, default=0)
    ap.add_argument('--opt', action='append', default=[], help='engine option name=value')
    ap.add_argument('--workload', default='rollout', choices=['rollout', 'mcts'],
                    help="'mcts' = BASELINE configs[2]: full lock-step MCTS (50 expansions, 10 samples, sim depth 5) over 64 episodes/GPU (secondary metric)")
    ap.add_argument('--episodes', type=int, default=64)
    ap.add_argument('--force-dist', action='store_true', help='initialise torch.distributed (RCCL) even with one rank: exercises the N>1 code path on a 1-GPU box')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-prof', action='store_true')
    a = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    use_dist = world > 1 or a.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    assert world == a.gpus, f'--gpus {a.gpus} but WORLD_SIZE={world}'
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)

    import daimc_amd
    R, S, D = a.rows, a.samples, a.depth
    model = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device=device, seed=1, row_offset=rank * R)
    if a.dec_chunk:
        model.set_option('dec_chunk', a.dec_chunk)
    for kv in a.opt:
        k_, v_ = kv.split('=')
        model.set_option(k_, int(v_))
    if a.workload == 'mcts':
        return bench_mcts(a, model, device, world, rank, dist)
    frames = synth_frames(R // 4, device, seed=100 + rank)
    o = frames.repeat_interleave(4, dim=0).contiguous()          # row 4i+a = (root i, action a), util.py:56-60
    pi = torch.eye(4, device=device).repeat(R // 4, 1).contiguous()

    def step(k):
        G, _, _ = model.calculate_G_repeated(o, pi, steps=D, samples=S, stage=k * D)
        P, _ = model.action_posterior(G)
        if use_dist:
            daimc_amd.gather_action_posteriors(P, world * (R // 4))      # the only RCCL traffic: [R/4, 4] floats per rank
        return G

    print(f'[bench] rank {rank}: model ready, warm-up', file=sys.stderr, flush=True)
    for k in range(a.warmup):
        step(k)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    DOM = 'dec_b_convT3_final_reduce'
    if not a.no_prof:
        model.prof_enable(True, classes=[DOM])     # HIP events around the dominant kernel only (3 launches per step)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        G = step(a.warmup + k)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    print(f'[bench] rank {rank}: timed region {dt:.3f}s', file=sys.stderr, flush=True)
    assert torch.isfinite(G).all()
    prof = model.prof_read() if not a.no_prof else {}
    breakdown = {}
    NB = 2
    if not a.no_prof:
        # per-class breakdown from extra, un-timed steps, ONE class at a time: event pairs around every launch of a step slow
        # all of its kernels down by ~10 % (the sum no longer matched ms_per_step)
        kk = a.warmup + a.steps
        for c in model.PROF_CLASSES:
            if c.startswith('unused'):
                continue
            model.prof_enable(True, classes=[c])
            for _ in range(NB):
                step(kk); kk += 1
            ms, n = model.prof_read()[c]
            breakdown[c] = (ms / NB, n // NB)
    model.prof_enable(False)
    if use_dist:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        value = world * R * a.steps / dt
        out = {
            'metric': 'EFE rollouts/sec (64x64 dSprites, 10 MC-samples, depth 5)', 'value': value, 'unit': 'rollouts/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'calculate_G_repeated: {R} rows ({R // 4} roots x 4 actions) x depth {D} x {S} MC samples per GPU '
                                   f'(BASELINE configs[1]) + action posterior' + (' + all_gather of posteriors' if world > 1 else ''),
                       'rows_per_gpu': R, 'samples': S, 'depth': D, 'parallelism': f'episodes sharded x{world}, weights replicated'},
            'achieved_tflops_total': value * 2 * MAC_ROLLOUT / 1e12,
        }
        if prof:
            name = 'dec_b_convT3_final_reduce'
            ms, n = prof[name]
            rows_per_launch = (a.steps * D * 3 * S * R) / max(n, 1)
            ach = (2 * MAC_DECB_ROW * rows_per_launch) / (ms / max(n, 1) * 1e-3) / 1e12 if ms > 0 else 0.0
            out['roofline'] = {'bound': 'mfma', 'kernel': 'k_dec_b (ConvTranspose2d 64->32 s2 + ConvTranspose2d 32->1 + sigmoid + per-image reduction, fused)',
                               'achieved': ach, 'peak': PEAK_FP32_MFMA_TF, 'unit': 'TFLOP/s', 'frac': ach / PEAK_FP32_MFMA_TF,
                               'traffic': None, 'launches': int(n), 'avg_launch_ms': ms / max(n, 1),
                               'flops_per_launch': 2 * MAC_DECB_ROW * rows_per_launch}
            # HBM bytes of the dominant kernel from the committed PMC profile (tools/gpu_profile.sh -> profiles/*_summary.txt)
            try:
                import glob
                import re
                best = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*rocprof_summary.txt')))[-1]
                m_ = re.search(r'== HBM traffic \(JSON\) ==\n(\{.*\})', open(best).read())
                tj = json.loads(m_.group(1))['k_dec_b']
                out['roofline']['traffic'] = (tj['hbm_read_bytes_per_image'] + tj['hbm_write_bytes_per_image']) * rows_per_launch
                out['roofline']['traffic_source'] = os.path.basename(best) + ' (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, per image x images per launch)'
            except Exception:
                pass
            tot = sum(v[0] for v in breakdown.values())
            kern = {}
            for k_, (ms_, n_) in breakdown.items():
                if n_ == 0:
                    continue
                e = {'ms': round(ms_, 3), 'launches': int(n_), 'share': round(ms_ / tot, 4) if tot else 0}
                if k_ in CLASS_MACS_PER_ROW and ms_ > 0:
                    e['tflops'] = round(2 * CLASS_MACS_PER_ROW[k_] * D * 3 * S * R / (ms_ * 1e-3) / 1e12, 2)
                kern[k_] = e
            out['kernels_one_step'] = kern
        if world == 1 and not a.no_cpu:
            out['cpu_baseline'] = cpu_baseline(D, S)
            out['speedup_vs_cpu_baseline'] = value / out['cpu_baseline']['value']
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
