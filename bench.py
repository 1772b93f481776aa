#!/usr/bin/env python
"""EFE rollouts/sec on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic input: `calculate_G_repeated` over R rows (R/4 root frames
x 4 actions) with 10 MC samples and depth 5, followed by the action posterior.
  * every N: R = 128 rows PER GPU (BASELINE configs[1], the configuration the metric is quoted on; 32 episodes x 4 actions),
    so the N = 1 point of a scaling curve is the BENCH line and every point runs the same per-GPU workload (weak scaling).
    Episodes shard with no data-path collective; at N > 1 ONE all_gather of the [32, 4] action posteriors per step is the
    only RCCL traffic (its own time is reported as `all_gather_ms`).
Inputs are resident in HBM before the timed region.  The timed region is exactly K steps between barrier + synchronize
pairs, with NO profiling inside it; it is REPEATED (same K) until about 10 s of GPU time have been measured, and
`ms_per_step` / `value` come from the median region (all region times are in the JSON), so the run is long enough to be
observed from outside.  Per-kernel HIP-event times (the `roofline` entry included) come from extra, un-timed steps after it.

The same JSON line carries, under "extras", the other BASELINE configurations: configs[2] (full lock-step MCTS, 64 episodes
per GPU x 50 expansions x 10 samples, simulation depth 5; at N > 1 this is configs[3]: the root visit distributions of
64 x N episodes are all-gathered) and configs[4] (3 x 84 x 84, 30 samples, depth 7, 32 episodes per GPU), each with its own
roofline and -- at N = 1 -- CPU baseline.

  python bench.py --gpus N --steps 20 --warmup 5          (N > 1 without WORLD_SIZE in the environment: bench.py launches its own
                                                           N ranks, one per GPU, through torch.distributed.run on 127.0.0.1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import glob
import json
import os
import re
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAC_TRANS, MAC_DEC, MAC_ENC, MAC_HABIT = 541_696, 43_256_320, 3_868_960, 18_176      # per network row
MAC_ROLLOUT = 6_739_934_560          # 51 encoder + 100 transition + 150 decoder passes
MAC_DECB_ROW = 18_874_368 + 1_179_648   # k_dec_b: ConvTranspose2d(64,32,3,s2) 32*32*9*64*32 + ConvTranspose2d(32,1,3,s1) 64*64*9*32
ALG_BYTES_DECB_IMAGE = 262144 + 4 + 16384 / 3   # k_dec_b per image: y2 read, one sum written, every third image (the D1 pass) stored
PEAK_FP32_MFMA_TF = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, fp32 in / fp32 acc
DOM = 'dec_b_convT3_final_reduce'


def class_macs_per_step(R, D, S):
    """algorithmic MACs of one rollout step, by kernel class (rows of each network x MACs per row)"""
    dec_rows, enc_rows, tr_rows = D * 3 * S * R, D * S * R + R, D * 2 * S * R
    return {
        'transition_mlp': tr_rows * MAC_TRANS,
        'dec_dense_small': dec_rows * (10 * 256 + 2 * 256 * 256),
        'dec_dense_16384': dec_rows * 256 * 16384,
        'dec_a_convT1_convT2': dec_rows * 2 * 256 * 9 * 64 * 64,
        DOM: dec_rows * MAC_DECB_ROW,
        'encoder': enc_rows * MAC_ENC,
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


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline(depth, samples, budget_s=14.0):
    """The oracle (CPU restatement of the reference's own torch op sequence, torch RNG like the reference) timed on this
    host's cores on a bounded sample of the same workload: >= 3 timed repeats on all usable cores (median + spread) and one
    1-thread figure.  BASELINE.md section 4 records the restatement-vs-reference time cross-check."""
    from oracle import synth
    from oracle.efe_oracle import OracleModel, TorchNoise
    cores = usable_cores()
    m = OracleModel(synth.make_weights(1234, 1.15), TorchNoise())

    def run(rows, d, s):
        o = torch.from_numpy(np.repeat(synth.make_frames(5, (rows + 3) // 4), 4, axis=0)[:rows])
        pi = torch.eye(4).repeat((rows + 3) // 4, 1)[:rows]
        t = time.perf_counter()
        with torch.no_grad():
            m.calculate_G_repeated(o, pi, d, False, s, 0)
        return time.perf_counter() - t

    torch.set_num_threads(cores)
    run(4, 1, 1)                                   # warm-up (thread pools, oneDNN primitives)
    t_unit = run(8, 1, 1)                          # 8 rows x 1 stage x 1 sample
    per_row_full = t_unit / 8 * depth * samples    # estimated seconds per full rollout row
    reps = 3
    rows = int(min(128, max(4, budget_s / reps / max(per_row_full, 1e-4))))
    rows -= rows % 4
    print(f'[bench] cpu_baseline: {cores} threads, unit pass {t_unit:.3f}s, timing {reps} x {rows} rows', file=sys.stderr, flush=True)
    times = [run(rows, depth, samples) for _ in range(reps)]
    rates = sorted(rows / t for t in times)
    torch.set_num_threads(1)
    t1 = run(4, depth, samples)                    # one root (4 rows) on ONE thread
    torch.set_num_threads(cores)
    return {'value': statistics.median(rates), 'unit': 'rollouts/s', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model(),
            'repeats': reps, 'min': rates[0], 'max': rates[-1], 'one_thread_value': 4 / t1,
            'sample': f'{reps} timed passes of {rows} rows x depth {depth} x {samples} MC samples after warm-up (median; '
                      f'{sum(times):.1f} s), plus 4 rows on 1 thread ({t1:.1f} s); torch-CPU eager oracle '
                      f'(oracle/efe_oracle.py, torch RNG like the reference)'}


def mcts_flops_per_decision(samples, repeats=50, depth=5):
    """algorithmic FLOPs of one planning decision: (repeats + 1) expansions of 4 rows x calculate_G(S) + repeats simulations
    (depth habit + transition steps, then a depth-row single-sample trajectory G) -- SURVEY 8a-12"""
    g_row = samples * (2 * MAC_TRANS + 3 * MAC_DEC + MAC_ENC)
    sim = depth * (MAC_HABIT + MAC_TRANS) + depth * (MAC_TRANS + 3 * MAC_DEC + MAC_ENC)
    return 2.0 * ((repeats + 1) * 4 * g_row + repeats * sim)


def cpu_baseline_mcts(samples, depth=5, repeats_sample=12, reps=3):
    """BASELINE configs[2] on the CPU as the reference would run it: ONE episode at a time through the oracle planner
    (oracle/mcts_oracle.py), torch RNG.  Bounded sample: `reps` timed passes (median) of `repeats_sample` planner iterations
    (+ the root expansion) instead of 50; decisions/s is extrapolated by algorithmic FLOPs (every iteration costs the same)."""
    from oracle import synth
    from oracle import mcts_oracle as MO
    from oracle.efe_oracle import OracleModel, TorchNoise
    cores = usable_cores()
    torch.set_num_threads(cores)
    m = OracleModel(synth.make_weights(1234, 1.15), TorchNoise())
    p = MO.Params(repeats=repeats_sample, simulation_depth=depth, use_means=False, threshold=2.0, samples=samples)
    frames = synth.make_frames(6, reps)
    MO.plan(m, torch.from_numpy(frames[0]), MO.Params(repeats=1, simulation_depth=2, use_means=False, threshold=2.0, samples=1), 0)     # warm-up
    times = []
    for r in range(reps):
        t = time.perf_counter()
        MO.plan(m, torch.from_numpy(frames[r]), p, 0)
        times.append(time.perf_counter() - t)
    dt = statistics.median(times)
    frac = mcts_flops_per_decision(samples, repeats_sample, depth) / mcts_flops_per_decision(samples, 50, depth)
    return {'value': frac / dt, 'unit': 'decisions/s', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model(), 'repeats': reps,
            'min': frac / max(times), 'max': frac / min(times),
            'sample': f'{reps} timed passes (median {dt:.1f} s; {sum(times):.1f} s in all) of one episode each, {repeats_sample} of 50 planner '
                      f'iterations (+ root expansion), {samples} MC samples, simulation depth {depth}, extrapolated to a full decision by '
                      f'algorithmic FLOPs (x{1 / frac:.1f}); sequential single-episode oracle planner (oracle/mcts_oracle.py over '
                      f'oracle/efe_oracle.py), torch RNG like the reference'}


def natural_key(path):
    return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', os.path.basename(path))]


def committed_traffic(kernel='k_dec_b', geometry='dsprites'):
    """HBM bytes per image of the dominant kernel from the newest committed PMC profile OF THE SAME GEOMETRY: the dSprites headline
    reads profiles/rN_vM_rocprof_summary.txt, the 3 x 84 x 84 leg profiles/rN_vM_ai_rocprof_summary.txt (both files key their
    dominant kernel 'k_dec_b'; the experiment profiles rN_vM_b3_* are never used).  Newest = highest (round, version) in natural
    order.  -> (bytes_per_image, file name) or (None, None)"""
    pat = {'dsprites': r'^r\d+_v\d+_rocprof_summary\.txt$', 'animalai': r'^r\d+_v\d+_ai_rocprof_summary\.txt$'}[geometry]
    files = sorted((f for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_rocprof_summary.txt')) if re.match(pat, os.path.basename(f))),
                   key=natural_key)
    for best in reversed(files):
        m_ = re.search(r'== HBM traffic \(JSON\) ==\n(\{.*\})', open(best).read())
        if not m_:
            continue
        tj = json.loads(m_.group(1)).get(kernel)
        if tj:
            return tj['hbm_read_bytes_per_image'] + tj['hbm_write_bytes_per_image'], os.path.basename(best)
    return None, None


def committed_split_traffic(kernel_prefix, opt):
    """HBM bytes per launch of a persistent split-operand kernel from the newest committed profile of that mode (profiles/rN_vM_b3_* for
    mfma_bf16x3, rN_vM_f16_* for mfma_f16x2; the profiled launch is the headline's 19 200 images).  -> (bytes, file name) or (None, None)"""
    tag = {'mfma_bf16x3': 'b3', 'mfma_f16x2': 'f16'}[opt]
    files = sorted((f for f in glob.glob(os.path.join(ROOT, 'profiles', f'r*_{tag}_rocprof_summary.txt'))
                    if re.match(rf'^r\d+_v\d+_{tag}_rocprof_summary\.txt$', os.path.basename(f))), key=natural_key)
    for best in reversed(files):
        m_ = re.search(r'== HBM traffic \(JSON\) ==\n(\{.*\})', open(best).read())
        if not m_:
            continue
        for k, tj in json.loads(m_.group(1)).items():
            if k.startswith(kernel_prefix) and 'hbm_read_bytes_per_launch' in tj:
                return tj['hbm_read_bytes_per_launch'] + tj['hbm_write_bytes_per_launch'], os.path.basename(best)
    return None, None


def timed_regions(step, steps, k0, rk, min_total_s=10.0, max_regions=60):
    """time EXACTLY `steps` steps between sync() pairs (barrier + device synchronise on both sides); repeat the region until
    min_total_s of measured time.  The stop decision uses the MAX-over-ranks time of every region (one scalar all_reduce outside
    the timed section), so that every rank runs the same number of regions -- a rank-local decision would leave one rank in a
    barrier that the others never enter.
    -> (region seconds, max over ranks), (this rank's region seconds), next step index"""
    out, local, k = [], [], k0
    while True:
        rk.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(k); k += 1
        rk.sync()
        dt = time.perf_counter() - t0
        local.append(dt)
        out.append(rk.max_over_ranks([dt])[0])
        if sum(out) >= min_total_s or len(out) >= max_regions:
            return out, local, k


def rank_of(rk):
    return int(getattr(rk, 'rank', 0))


class ClockSampler:
    """Shader clock of this rank's GPU while the timed regions run, sampled from a SIDE THREAD that only reads the driver's sysfs file
    (/sys/class/drm/card*/device/pp_dpm_sclk: the level marked '*'; rocm-smi --showclocks as a fallback) -- nothing is put on any HIP
    stream.  The fp32 MFMA peak of the roofline entries assumes 2.4 GHz (MI355X_MICROARCH.md); boards in this pool hold 2.2 - 2.4 GHz under
    this load (power), so the same binary scores 0.826 - 0.85 by lease: `frac_at_measured_clock` = frac x 2400 / sclk separates a kernel
    regression from a slow board."""
    NOMINAL_MHZ = 2400.0

    def __init__(self, device, period=0.1):
        import threading
        self.period, self.samples, self._stop = period, [], threading.Event()
        self.path, self.index = None, device.index or 0
        try:
            import glob
            pr = torch.cuda.get_device_properties(device)
            want = None
            if hasattr(pr, 'pci_bus_id'):
                want = '%04x:%02x:%02x' % (getattr(pr, 'pci_domain_id', 0), pr.pci_bus_id, getattr(pr, 'pci_device_id', 0))
            cards = []
            for f in sorted(glob.glob('/sys/class/drm/card[0-9]*/device/pp_dpm_sclk')):
                real = os.path.realpath(os.path.dirname(f))
                if os.path.exists(os.path.join(os.path.dirname(f), 'vendor')) and open(os.path.join(os.path.dirname(f), 'vendor')).read().strip() != '0x1002':
                    continue
                cards.append((real, f))
            match = [f for real, f in cards if want and want in real]
            self.path = match[0] if match else (cards[self.index][1] if self.index < len(cards) else None)
        except Exception:
            self.path = None
        self.source = 'sysfs pp_dpm_sclk' if self.path else 'rocm-smi --showclocks'
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        import re
        if self.path:
            for line in open(self.path).read().splitlines():
                if line.rstrip().endswith('*'):
                    m = re.search(r'(\d+)\s*Mhz', line, re.I)
                    return float(m.group(1)) if m else None
            return None
        import subprocess
        r = subprocess.run(['/opt/rocm/bin/rocm-smi', '-d', str(self.index), '--showclocks'], capture_output=True, text=True, timeout=5)
        m = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', r.stdout)
        return float(m.group(1)) if m else None

    def _run(self):
        while not self._stop.is_set():
            try:
                v = self._read()
                if v:
                    self.samples.append(v)
            except Exception:
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=6)

    def report(self, frac=None):
        if not self.samples:
            return {'sclk_mhz_under_load': None, 'frac_at_measured_clock': None, 'sclk_source': self.source + ' (no sample)'}
        xs = sorted(self.samples)
        med = xs[len(xs) // 2]
        out = {'sclk_mhz_under_load': med, 'sclk_mhz_min': xs[0], 'sclk_mhz_max': xs[-1], 'sclk_samples': len(xs), 'sclk_source': self.source,
               'sclk_nominal_mhz': self.NOMINAL_MHZ}
        if frac is not None and med > 0:
            out['frac_at_measured_clock'] = frac * self.NOMINAL_MHZ / med
        return out


class Ranks:
    """the process group of this run (None at one rank without --force-dist): barrier, max-over-ranks of the region times,
    per-rank figures, and the one data-path collective (all_gather of the action posteriors)"""

    def __init__(self, dist, world, rank, device, backend):
        self.dist, self.world, self.rank, self.device, self.backend = dist, world, rank, device, backend
        self.on = dist is not None
        self.red_dev = device if backend == 'nccl' else torch.device('cpu')     # gloo (the --share-device launcher test): host tensors

    def sync(self):
        torch.cuda.synchronize()
        if self.on:
            self.dist.barrier()

    def max_over_ranks(self, xs):
        if not self.on:
            return [float(x) for x in xs]
        t = torch.tensor(xs, device=self.red_dev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in t]

    def per_rank(self, x):
        if not self.on:
            return [float(x)]
        out = torch.zeros(self.world, device=self.red_dev, dtype=torch.float64)
        self.dist.all_gather_into_tensor(out, torch.tensor([x], device=self.red_dev, dtype=torch.float64))
        return [float(v) for v in out]

    def gather(self, P, n_total):
        import daimc_amd
        if not self.on:
            return P
        return daimc_amd.gather_action_posteriors(P if self.backend == 'nccl' else P.cpu(), n_total)

    def time_gather(self, P, n_total, iters=50):
        """host-observed milliseconds per all_gather of the action posteriors, back to back (max over ranks)"""
        if not self.on:
            return None
        for _ in range(5):
            self.gather(P, n_total)
        self.sync()
        t0 = time.perf_counter()
        for _ in range(iters):
            self.gather(P, n_total)
        torch.cuda.synchronize()
        return self.max_over_ranks([1e3 * (time.perf_counter() - t0) / iters])[0]

    def info(self):
        if not self.on:
            return {}
        # (rccl_ranks only when the group really is RCCL: the one-device launcher rehearsal runs over gloo and says so)
        n, be = int(self.dist.get_world_size()), str(self.dist.get_backend())
        return {'rccl_ranks' if be == 'nccl' else 'ranks': n, 'backend': be}

    def check_devices(self, model, local, share_device):
        """rank r really runs on GPU r: the engine context's own device (efe_get_device) is the local rank's, and no two ranks of the job hold
        the same PCI device (unless --share-device asked for exactly that) -> the per-rank (device index, PCI bus id) list for the line"""
        idx, bus = model.engine_device()
        if idx != local or idx != torch.cuda.current_device():
            sys.exit(f'[bench] rank {self.rank}: engine context on device {idx}, expected local rank {local} (current device {torch.cuda.current_device()})')
        if not self.on:
            return [[idx, bus]]
        mine = [None] * self.world
        self.dist.all_gather_object(mine, [idx, bus])
        if not share_device and len({b for _, b in mine}) != self.world:
            sys.exit(f'[bench] {self.world} ranks on {len({b for _, b in mine})} distinct GPUs: {mine} -- one rank per GPU is required (or --share-device)')
        return mine


def bench_mcts(a, model, device, rk, steps, warmup, with_cpu, threshold=2.0, min_total_s=2.5):
    """BASELINE configs[2]/[3]: E episodes per GPU, each a full MCTS decision (50 expansions with S MC samples,
    simulation depth 5, use_means=False), planned in lock-step; the root visit distributions are gathered across ranks
    (N > 1: configs[3]).  threshold = 2.0 disables the early stop (every episode does all 50 expansions: the FLOP count of the
    roofline entry is exact); threshold = 0.5 is the reference's default (mcts.py:139) and times the device-side row mask."""
    import daimc_amd
    E, world, rank = a.episodes, rk.world, rk.rank
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.use_means, p.threshold, p.samples = 50, 5, False, threshold, a.samples
    frames = synth_frames(E, device, seed=200 + rank)
    last = {}

    work = []                                        # per decision batch: fraction of the 50 x E episode-iterations that ran

    def step(_k):
        out, distn = daimc_amd.active_inference_mcts_batch(model, frames, p, o_shape=(1, 64, 64), episode_offset=rank * E)
        work.append(sum(o[1] for o in out) / (50.0 * E))
        last['out'], last['P'] = out, distn.to(device)
        rk.gather(last['P'], world * E)
        return out

    for k in range(warmup):
        step(k)
    del work[:]
    with ClockSampler(device) as clk:
        regions, local, _ = timed_regions(step, steps, 0, rk, min_total_s=min_total_s, max_regions=12)
    work_timed = list(work)
    per_rank_ms = rk.per_rank(1e3 * statistics.median(local) / steps)
    dt = statistics.median(regions)
    dec = world * E * steps / dt
    iters = [o[1] for o in last['out']]
    out = {'metric': 'MCTS decisions/sec (50 expansions, %d MC samples, sim depth 5)' % a.samples, 'value': dec,
           'unit': 'decisions/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': 1e3 * dt / steps,
           'timed_regions': len(regions), 'region_ms': [round(1e3 * x, 3) for x in regions],
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': f'lock-step MCTS, {E} episodes per GPU x 50 expansions x {a.samples} MC samples, simulation depth 5, '
                                  f'early-stop threshold {threshold} (BASELINE configs[{2 if world == 1 else 3}])'
                                  + (' + all_gather of the root visit distributions' if rk.on else ''),
                      'episodes_per_gpu': E, 'threshold': threshold,
                      # the code path of this leg (device-side stop, lagged host check, compaction of stopped episodes) is pinned to the reference
                      # planner's own early stops at this depth and threshold by tests/golden/mcts_deep_s10_thr.npz (oracle/make_golden_thr.py;
                      # tests/test_gpu_parity.py::test_lockstep_batch_with_early_stops_contains_the_reference_episodes)
                      **({'parity_fixture': 'tests/golden/mcts_deep_s10_thr.npz (threshold 0.5: reference stops before iterations 41, 50, 50, 50, 29, 50)'}
                         if abs(threshold - 0.5) < 1e-9 else {})},
           'iterations_done_mean': float(np.mean(iters)), 'iterations_done_min': int(min(iters)),
           'work_fraction_mean': float(np.mean(work_timed)), 'work_fraction_per_decision_batch': [round(x, 4) for x in work_timed]}
    out.update(rk.info())
    out['devices'] = rk.check_devices(model, torch.cuda.current_device(), bool(getattr(a, 'share_device', False)))
    if rk.on:
        out['per_rank_ms_per_step'] = [round(x, 3) for x in per_rank_ms]
        out['all_gather_ms'] = rk.time_gather(last['P'], world * E)
    if threshold >= 0.75:        # no early stop: every decision is exactly 51 expansions + 50 simulations
        fl = mcts_flops_per_decision(a.samples)
        out.update({'gflop_per_decision': fl / 1e9, 'rollout_equivalents_per_s': dec * fl / (2 * MAC_ROLLOUT),
                    'achieved_tflops_total': dec * fl / 1e12})
        # whole-workload roofline (the planner is many launches; its dominant kernel is the same k_dec_b) + the dominant kernel
        # alone, timed with HIP events in ONE extra, un-timed decision batch
        out['roofline'] = {'bound': 'mfma', 'kernel': 'whole decision (all kernels); k_dec_b alone under "dominant"',
                           'achieved': dec * fl / 1e12 / world, 'peak': PEAK_FP32_MFMA_TF, 'unit': 'TFLOP/s',
                           'frac': dec * fl / 1e12 / world / PEAK_FP32_MFMA_TF, 'traffic': None}
        out['roofline'].update(clk.report(out['roofline']['frac']))
        if not a.no_prof:
            # the planner's simulations run on a replica context (second stream): time the class on both contexts
            ctxs = [model] + ([model._replica] if getattr(model, '_replica', None) is not None else [])
            for c_ in ctxs:
                c_.prof_enable(True, classes=[DOM])
            step(0)
            ms_dom = n_dom = 0
            for c_ in ctxs:
                ms_, n_ = c_.prof_read()[DOM]
                ms_dom += ms_; n_dom += n_
                c_.prof_enable(False)
            n_img = 51 * 3 * a.samples * 4 * E + 50 * 3 * 5 * E     # decoder images of one decision batch (expansions + simulations)
            ach_dom = 2 * MAC_DECB_ROW * n_img / (ms_dom * 1e-3) / 1e12 if ms_dom > 0 else 0.0
            out['roofline']['dominant'] = {'kernel': 'k_dec_b', 'achieved': ach_dom, 'frac': ach_dom / PEAK_FP32_MFMA_TF,
                                           'launches': int(n_dom), 'avg_launch_ms': ms_dom / max(n_dom, 1),
                                           'note': 'HIP events of this rank on both engine contexts (expansions on the main stream, simulations on '
                                                   'the replica stream: the two overlap, so a launch shares the GPU with the other stream\'s kernels), '
                                                   'one un-timed decision batch'}
            # the same launches with the GPU to themselves: one more un-timed decision batch planned WITHOUT the second stream
            # (MCTS_Params.overlap_simulate = False: simulations behind the expansions on the main stream), so a k_dec_b launch of the
            # planner's sizes (7680 images per expansion, 960 per simulation) is timed alone
            import copy
            p1 = copy.copy(p)
            p1.overlap_simulate = False
            model.prof_enable(True, classes=[DOM])
            daimc_amd.active_inference_mcts_batch(model, frames, p1, o_shape=(1, 64, 64), episode_offset=rank * E)
            ms1, n1 = model.prof_read()[DOM]
            model.prof_enable(False)
            if ms1 > 0:
                ach1 = 2 * MAC_DECB_ROW * n_img / (ms1 * 1e-3) / 1e12
                out['roofline']['dominant_unshared'] = {'kernel': 'k_dec_b', 'achieved': ach1, 'frac': ach1 / PEAK_FP32_MFMA_TF, 'launches': int(n1),
                                                        'avg_launch_ms': ms1 / max(n1, 1),
                                                        'note': 'one un-timed decision batch with the simulations on the main stream: 51 launches of '
                                                                '7680 images + 50 of 960, each with the GPU to itself'}
    if with_cpu and rank == 0:
        out['cpu_baseline'] = cpu_baseline_mcts(a.samples)
        out['speedup_vs_cpu_baseline'] = dec / out['cpu_baseline']['value']
    return out


def bench_single_episode(model, device, samples):
    """the reference's own call shape (/root/reference/src/mcts.py:150-195, 64-86): ONE episode -- active_inference_mcts with 4-row
    expansions and batch-1 simulations -- as host-observed latency: milliseconds per 50-iteration decision (use_means, S = 1; and the
    benchmark's S MC samples per expansion, through the lock-step planner at E = 1) and per
    calculate_G / calculate_G_mean call on the 4 action rows of one state.  Latency-bound (dependent launches of a few images each): a
    per-call figure, not a roofline entry."""
    import daimc_amd
    frame = synth_frames(1, device, seed=300)

    def ms(fn, n):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n
    s4 = torch.randn(4, 10, device=device) * 0.3
    out = {'unit': 'ms', 'what': 'one episode, host-observed latency per call (mean of repeated calls, device noise)',
           'calculate_G_mean_4rows': ms(lambda: model.calculate_G_mean(s4, model.pi_one_hot), 30),
           'calculate_G_4rows_S%d' % samples: ms(lambda: model.calculate_G(s4, model.pi_one_hot, samples=samples), 30),
           'mcts_step_simulate_depth5': ms(lambda: model.mcts_step_simulate(s4[0], 5), 30)}
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.threshold, p.use_means = 50, 5, 2.0, True
    out['decision_use_means_S1_reference_api'] = ms(lambda: daimc_amd.active_inference_mcts(model, frame[0], p, o_shape=(1, 64, 64)), 3)
    p.host_tree = True            # the host-side Node tree (what a caller of Node.expand / select / backpropagate drives)
    out['decision_use_means_S1_host_node_tree'] = ms(lambda: daimc_amd.active_inference_mcts(model, frame[0], p, o_shape=(1, 64, 64)), 3)
    q = daimc_amd.MCTS_Params()
    q.repeats, q.simulation_depth, q.threshold, q.use_means, q.samples = 50, 5, 2.0, False, samples
    key = 'decision_S%d_lockstep_E1_launched' % samples
    out[key] = ms(lambda: daimc_amd.active_inference_mcts_batch(model, frame, q, o_shape=(1, 64, 64)), 4)
    out[key.replace('decision', 'iteration')] = out[key] / 51.0          # root expansion + 50 iterations
    return out


def cpu_baseline_generic(A, C, R, depth, samples):
    """configs[4] on the CPU: the build-defined oracle restatement (parity unpinned), 3 bounded passes of >= 3 s each;
    rollouts/s extrapolated by (depth x samples) -- every (stage, sample) costs the same"""
    from oracle import synth
    from oracle.efe_oracle import OracleModel, TorchNoise
    cores = usable_cores()
    torch.set_num_threads(cores)
    m = OracleModel(synth.make_weights(1234, 1.15, A, C, R), TorchNoise(), pi_dim=A, channels=C, resolution=R)
    n_ep = 8
    o = torch.from_numpy(np.repeat(synth.make_frames_rgb(5, n_ep, C, R), A, axis=0))
    pi = torch.eye(A).repeat(n_ep, 1)
    times = []
    with torch.no_grad():
        m.calculate_G_repeated(o[:A], pi[:A], 1, False, 1, 0)
        t = time.perf_counter()
        m.calculate_G_repeated(o, pi, 1, False, 2, 0)
        unit = (time.perf_counter() - t) / 2            # seconds per (stage, sample) of the 24-row batch
        d = 3
        sm = int(min(samples, max(4, round(3.2 / max(unit, 1e-3) / d))))
        for rep in range(3):
            t = time.perf_counter()
            m.calculate_G_repeated(o, pi, d, False, sm, rep)
            times.append(time.perf_counter() - t)
    dt = statistics.median(times)
    rows = o.shape[0]
    frac = (d * sm) / (depth * samples)
    return {'value': rows * frac / dt, 'unit': 'rollouts/s', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model(), 'repeats': len(times),
            'min': rows * frac / max(times), 'max': rows * frac / min(times),
            'sample': f'3 timed passes of {rows} rows x depth {d} x {sm} MC samples (median {dt:.1f} s, {sum(times):.1f} s in all), extrapolated to '
                      f'depth {depth} x {samples} samples (x{1 / frac:.1f}: every (stage, sample) costs the same); build-defined oracle '
                      f'restatement (oracle/efe_oracle.py, channels={C}, resolution={R}), torch RNG; parity unpinned'}


def bench_generic(a, device, rk, steps, warmup, with_cpu, min_total_s=2.0):
    """BASELINE configs[4]: Animal-AI-sized observations (3 x 84 x 84, 3 actions), 30 MC samples, depth 7, 32 episodes per GPU
    (256 over 8 GPUs).  Build-defined network, PARITY UNPINNED: no reference semantics exist for this geometry."""
    import daimc_amd
    world, rank = rk.world, rk.rank
    A, C, R, S, D, E = 3, 3, 84, 30, 7, 32
    rows = E * A
    model = daimc_amd.ActiveInferenceModel(10, A, 0.0, 1.0, 1.0, colour_channels=C, resolution=R, device=device, seed=1, row_offset=rank * rows)
    for kv in a.opt:
        k_, v_ = kv.split('=')
        model.set_option(k_, int(v_))
    g = torch.Generator().manual_seed(300 + rank)
    frames = torch.rand(E, C, R, R, generator=g).to(device)
    o = frames.repeat_interleave(A, dim=0).contiguous()
    pi = torch.eye(A, device=device).repeat(E, 1).contiguous()
    model.reserve(rows, D, S)
    last = {}

    def step(k):
        G, _, _ = model.calculate_G_repeated(o, pi, steps=D, samples=S, stage=k * D)
        P, _ = model.action_posterior(G, A)
        last['P'] = P
        rk.gather(P, world * E)
        return G

    for k in range(warmup):
        step(k)
    rk.sync()
    macs_row = model.last_call_macs() / rows
    with ClockSampler(device) as clk:
        regions, local, _ = timed_regions(step, steps, warmup, rk, min_total_s=min_total_s, max_regions=12)
    per_rank_ms = rk.per_rank(1e3 * statistics.median(local) / steps)
    dt = statistics.median(regions)
    value = world * rows * steps / dt
    tf = value * 2 * macs_row / 1e12
    out = {'metric': 'EFE rollouts/sec (84x84 RGB, 30 MC-samples, depth 7)', 'value': value, 'unit': 'rollouts/s', 'n_gpus': world,
           'steps': steps, 'warmup': warmup, 'ms_per_step': 1e3 * dt / steps, 'timed_regions': len(regions),
           'region_ms': [round(1e3 * x, 3) for x in regions], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'f32', 'data': 'synthetic', 'parity': 'unpinned: build-defined network, no reference semantics (SURVEY 8a-13)',
           'config': {'workload': f'calculate_G_repeated: {rows} rows ({E} episodes x {A} actions) x depth {D} x {S} MC samples per GPU, '
                                  f'{C} x {R} x {R} observations (BASELINE configs[4]) + action posterior'
                                  + (' + all_gather of posteriors' if rk.on else ''), 'rows_per_gpu': rows},
           'gflop_per_rollout': 2 * macs_row / 1e9, 'achieved_tflops_total': tf,
           'roofline': {'bound': 'mfma', 'kernel': 'whole step (generic-geometry path)', 'achieved': tf / world,
                        'peak': PEAK_FP32_MFMA_TF, 'unit': 'TFLOP/s', 'frac': tf / world / PEAK_FP32_MFMA_TF, 'traffic': None}}
    out.update(rk.info())
    out['devices'] = rk.check_devices(model, torch.cuda.current_device(), bool(getattr(a, 'share_device', False)))
    if rk.on:
        out['per_rank_ms_per_step'] = [round(x, 3) for x in per_rank_ms]
        out['all_gather_ms'] = rk.time_gather(last['P'], world * E)
    if not a.no_prof:
        # per-class HIP-event times of ONE un-timed step each (events around every launch inflate a step, so one class per pass)
        Bq = R // 4
        imgs = D * 3 * S * rows                                     # decoder images per step
        macs = {'dec_dense_16384': 256 * 64 * Bq * Bq, 'convT1_generic': Bq * Bq * 9 * 64 * 64, 'dec_a_convT1_convT2': Bq * Bq * 9 * 64 * 64,
                'dec_b_convT3_final_reduce': 4 * Bq * Bq * 9 * 64 * 32, 'final_layer_generic': R * R * 9 * 32 * C}
        names = model.generic_class_names()
        kern, kk = {}, warmup + 100
        for cname in names:
            model.prof_enable(True, classes=[cname])
            step(kk); kk += 1
            ms, n = model.prof_read()[cname]
            if n == 0:
                continue
            e = {'kernel': names[cname], 'ms': round(ms, 3), 'launches': int(n)}
            mc = macs.get(cname, 0)
            if cname == 'dec_a_convT1_convT2' and 'convT1_generic' not in kern:
                mc += macs['convT1_generic']                        # no ConvT1 launch of its own: k_convt_12 ran both layers (one class)
                e['kernel'] = 'k_convt_12 (ConvT 64->64 s1 + ReLU + ConvT 64->64 s2 + ReLU, fused)'
            if cname == 'dec_b_convT3_final_reduce' and 'final_layer_generic' not in names:
                mc += macs['final_layer_generic']                   # fused ConvT3 + final layer: one class
            if mc and ms > 0:
                e['tflops'] = round(2 * mc * imgs / (ms * 1e-3) / 1e12, 2)
                e['frac_of_fp32_mfma_peak'] = round(e['tflops'] / PEAK_FP32_MFMA_TF, 4)
            kern[cname] = e
        model.prof_enable(False)
        out['kernels_one_step'] = kern
        dom = kern['dec_b_convT3_final_reduce']
        out['roofline'] = {'bound': 'mfma', 'kernel': dom['kernel'] + ' (largest class of the step)',
                           'achieved': dom.get('tflops', 0.0), 'peak': PEAK_FP32_MFMA_TF, 'unit': 'TFLOP/s',
                           'frac': dom.get('frac_of_fp32_mfma_peak', 0.0), 'traffic': None, 'launches': dom['launches'],
                           'avg_launch_ms': dom['ms'] / max(dom['launches'], 1), 'whole_step_frac': tf / world / PEAK_FP32_MFMA_TF}
    out['roofline'].update(clk.report(out['roofline']['frac']))
    if 'launches' in out['roofline'] and (C, R) == (3, 84):
        bpi, src = committed_traffic('k_dec_b', 'animalai')       # the 3 x 84 x 84 profile (tools/gpu_profile.sh <tag> --workload animalai)
        if bpi is not None:
            n_l = max(out['roofline']['launches'], 1)
            out['roofline']['traffic'] = bpi * imgs / n_l
            out['roofline']['traffic_unit'] = 'bytes per launch'
            out['roofline']['traffic_from_profile'] = {'bytes_per_image': bpi, 'images_per_launch': imgs / n_l, 'file': 'profiles/' + src,
                                                       'how': 'rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE in separate passes, per image x images per launch'}
    if with_cpu and rank_of(rk) == 0:
        out['cpu_baseline'] = cpu_baseline_generic(A, C, R, D, S)
        out['speedup_vs_cpu_baseline'] = value / out['cpu_baseline']['value']
    return out


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start N ranks (one per GPU) of this script through
    torch.distributed.run on 127.0.0.1 and pass rank 0's JSON line through."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < a.gpus and not a.share_device:
        print(f'[bench] --gpus {a.gpus} but only {n_dev} HIP device(s) are visible (one rank per GPU; --share-device puts every rank '
              f'on cuda:0 over gloo to test the launcher on a 1-GPU box)', file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('[bench] launching ' + ' '.join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--rows', type=int, default=0, help='rollout rows per GPU (default 128 = BASELINE configs[1] on every GPU)')
    ap.add_argument('--samples', type=int, default=10)
    ap.add_argument('--depth', type=int, default=5)
    ap.add_argument('--dec-chunk', type=int, default=0)
    ap.add_argument('--opt', action='append', default=[], help='engine option name=value')
    ap.add_argument('--workload', default='rollout', choices=['rollout', 'mcts', 'animalai'],
                    help="'mcts' = only BASELINE configs[2]/[3]: full lock-step MCTS (50 expansions, 10 samples, sim depth 5) over 64 episodes/GPU; "
                         "'animalai' = only BASELINE configs[4]: 3 x 84 x 84 observations, 30 samples, depth 7, 32 episodes/GPU (parity unpinned)")
    ap.add_argument('--episodes', type=int, default=64)
    ap.add_argument('--threshold', type=float, default=2.0, help="--workload mcts: early-stop threshold (2.0 = disabled, 0.5 = the reference's default)")
    ap.add_argument('--force-dist', action='store_true', help='initialise torch.distributed (RCCL) even with one rank: exercises the N>1 code path on a 1-GPU box')
    ap.add_argument('--share-device', action='store_true', help='N > 1 on a box with fewer GPUs: every rank uses cuda:0 and the gather runs over gloo (launcher test only)')
    ap.add_argument('--min-seconds', type=float, default=10.0, help='measured GPU time of the headline workload (the K-step region is repeated)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-prof', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the configs[2]/[3] (MCTS) and configs[4] legs')
    ap.add_argument('--single-region', action='store_true', help='time the K steps once (no repeats)')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(a))
    # stdout carries exactly ONE line, the JSON: everything else that writes to file descriptor 1 (RCCL prints a version banner there
    # when the process group is torn down) is pointed at stderr; the line itself goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = 0 if a.share_device else int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    backend = None
    use_dist = world > 1 or a.force_dist
    if world != a.gpus:
        sys.exit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = 'gloo' if a.share_device else 'nccl'
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)       # one device per rank; "nccl" is RCCL on ROCm
        else:
            dist.init_process_group('gloo')
    rk = Ranks(dist, world, rank, device, backend)
    solo = world == 1 and not use_dist
    with_cpu = not a.no_cpu                    # N > 1: rank 0 measures the headline's CPU leg after the timed regions (the extras' CPU legs are N = 1 only)

    def emit(out):
        if rank == 0:
            os.write(real_stdout, (json.dumps(out) + '\n').encode())
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()

    import daimc_amd
    if a.workload == 'animalai':
        return emit(bench_generic(a, device, rk, max(1, min(a.steps, 3)), max(1, min(a.warmup, 1)), with_cpu, min_total_s=a.min_seconds))
    R = a.rows or 128
    S, D = a.samples, a.depth
    model = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device=device, seed=1, row_offset=rank * R)
    if a.dec_chunk:
        model.set_option('dec_chunk', a.dec_chunk)
    for kv in a.opt:
        k_, v_ = kv.split('=')
        model.set_option(k_, int(v_))
    if a.workload == 'mcts':
        return emit(bench_mcts(a, model, device, rk, max(1, min(a.steps, 3)), max(1, min(a.warmup, 1)), with_cpu, threshold=a.threshold,
                               min_total_s=a.min_seconds))
    model.reserve(R, D, S)                                       # steady-state steps never hipMalloc
    frames = synth_frames(R // 4, device, seed=100 + rank)
    o = frames.repeat_interleave(4, dim=0).contiguous()          # row 4i+a = (root i, action a), util.py:56-60
    pi = torch.eye(4, device=device).repeat(R // 4, 1).contiguous()
    last = {}

    def step(k):
        G, _, _ = model.calculate_G_repeated(o, pi, steps=D, samples=S, stage=k * D)
        P, _ = model.action_posterior(G)
        last['P'] = P
        rk.gather(P, world * (R // 4))             # the only RCCL traffic: [R/4, 4] floats per rank
        return G

    print(f'[bench] rank {rank}: model ready, warm-up', file=sys.stderr, flush=True)
    for k in range(a.warmup):
        step(k)
    rk.sync()
    grows0 = model.arena_stats()['grow_count']
    with ClockSampler(device) as clk:
        regions, local, kk = timed_regions(step, a.steps, a.warmup, rk, min_total_s=0.0 if a.single_region else a.min_seconds)
    G = step(kk); kk += 1
    torch.cuda.synchronize()
    print(f'[bench] rank {rank}: {len(regions)} timed regions of {a.steps} steps, {sum(regions):.3f}s', file=sys.stderr, flush=True)
    assert torch.isfinite(G).all()
    assert model.arena_stats()['grow_count'] == grows0, 'the scratch arena grew inside the timed region'
    per_rank_ms = rk.per_rank(1e3 * statistics.median(local) / a.steps)
    dt = statistics.median(regions)                  # per region: the slowest rank (max over ranks)
    gather_ms = rk.time_gather(last['P'], world * (R // 4))
    breakdown = {}
    if not a.no_prof:
        # per-class HIP-event times from extra, UN-TIMED steps, one class at a time: event pairs around every launch of a step
        # slow all of its kernels down by ~10 % (the sum no longer matched ms_per_step).  The dominant class gets more steps.
        for c in model.PROF_CLASSES:
            if c.endswith('_generic'):            # classes of the generic-geometry path only
                continue
            nb = 10 if c == DOM else 3
            model.prof_enable(True, classes=[c])
            for _ in range(nb):
                step(kk); kk += 1
            ms, n = model.prof_read()[c]
            breakdown[c] = (ms / nb, n // nb, ms, n)
        model.prof_enable(False)

    devs = rk.check_devices(model, device.index, a.share_device)           # every rank (a collective at N > 1)
    out = None
    if rank == 0:
        value = world * R * a.steps / dt
        cfg = 'BASELINE configs[1]' + (' on every GPU' if world > 1 else '') if R == 128 else 'custom size'
        out = {
            'metric': 'EFE rollouts/sec (64x64 dSprites, 10 MC-samples, depth 5)', 'value': value, 'unit': 'rollouts/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'calculate_G_repeated: {R} rows ({R // 4} roots x 4 actions) x depth {D} x {S} MC samples per GPU '
                                   f'({cfg}) + action posterior' + (' + all_gather of posteriors' if use_dist else ''),
                       'rows_per_gpu': R, 'samples': S, 'depth': D, 'parallelism': f'episodes sharded x{world}, weights replicated'},
            'timed_regions': len(regions), 'region_ms': [round(1e3 * x, 3) for x in regions],
            'ms_per_step_min': 1e3 * min(regions) / a.steps, 'ms_per_step_max': 1e3 * max(regions) / a.steps,
            'achieved_tflops_total': value * 2 * MAC_ROLLOUT / 1e12,
            'frac_of_fp32_mfma_peak_whole_step': value * 2 * MAC_ROLLOUT / 1e12 / world / PEAK_FP32_MFMA_TF,
        }
        out.update(rk.info())
        out['devices'] = devs
        if use_dist:
            out['per_rank_ms_per_step'] = [round(x, 3) for x in per_rank_ms]
            out['all_gather_ms'] = gather_ms
            if a.share_device:
                out['share_device'] = True
        if breakdown:
            _, _, ms, n = breakdown[DOM]
            rows_per_launch = D * 3 * S * R / max(1, n // 10)     # one k_dec_b launch per step when dec_chunk >= rows
            ach = (2 * MAC_DECB_ROW * rows_per_launch) / (ms / max(n, 1) * 1e-3) / 1e12 if ms > 0 else 0.0
            out['roofline'] = {'bound': 'mfma', 'kernel': 'k_dec_b (ConvTranspose2d 64->32 s2 + ConvTranspose2d 32->1 + sigmoid + per-image reduction, fused)',
                               'achieved': ach, 'peak': PEAK_FP32_MFMA_TF, 'unit': 'TFLOP/s', 'frac': ach / PEAK_FP32_MFMA_TF,
                               'traffic': None, 'launches': int(n), 'avg_launch_ms': ms / max(n, 1),
                               'flops_per_launch': 2 * MAC_DECB_ROW * rows_per_launch,
                               'timing': 'HIP events on the launch stream around the kernel, 10 un-timed steps after the timed regions'}
            out['roofline'].update(clk.report(ach / PEAK_FP32_MFMA_TF))
            # traffic = HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE in their own rocprofv3 passes, corrected as
            # MI355X_MICROARCH.md prescribes).  Counters need rocprofv3 around the process, so the figure is the one of the newest COMMITTED
            # profile of the same kernel and geometry (tools/gpu_profile.sh + tools/prof_summary.py), per image x the images of this launch
            bpi, src = committed_traffic('k_dec_b', 'dsprites')
            if bpi is not None:
                out['roofline']['traffic'] = bpi * rows_per_launch
                out['roofline']['traffic_unit'] = 'bytes per launch'
                out['roofline']['traffic_over_algorithmic'] = bpi / ALG_BYTES_DECB_IMAGE
                out['roofline']['traffic_from_profile'] = {'bytes_per_image': bpi, 'images_per_launch': rows_per_launch, 'file': 'profiles/' + src,
                                                           'how': 'rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE in separate passes, per image x images per launch'}
            tot = sum(v[0] for v in breakdown.values())
            macs = class_macs_per_step(R, D, S)
            kern = {}
            for k_, (ms_, n_, _, _) in breakdown.items():
                if n_ == 0:
                    continue
                e = {'ms': round(ms_, 3), 'launches': int(n_), 'share': round(ms_ / tot, 4) if tot else 0}
                if k_ in macs and ms_ > 0:
                    e['tflops'] = round(2 * macs[k_] / (ms_ * 1e-3) / 1e12, 2)
                    e['frac_of_fp32_mfma_peak'] = round(e['tflops'] / PEAK_FP32_MFMA_TF, 4)
                kern[k_] = e
            out['kernels_one_step'] = kern
        if with_cpu:
            out['cpu_baseline'] = cpu_baseline(D, S)
            out['speedup_vs_cpu_baseline'] = value / out['cpu_baseline']['value']
    pipelined = None
    if not a.no_extras and solo:
        # the same steps alternated over TWO engine contexts on two HIP streams (one context per stream, include/efe_engine.h): step k + 1's
        # sequential transition stages and small launches run beside step k's decoder kernels.  Reported beside the headline, not as it.
        m2 = model.replica()
        m2.reserve(R, D, S)
        ss = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]

        def step2(k):
            mm, st_ = (model, ss[0]) if k % 2 == 0 else (m2, ss[1])
            with torch.cuda.stream(st_):
                G2, _, _ = mm.calculate_G_repeated(o, pi, steps=D, samples=S, stage=k * D)
                mm.action_posterior(G2)
        for k in range(4):
            step2(k)
        regs2, _, kk = timed_regions(step2, a.steps, kk, rk, min_total_s=2.0, max_regions=12)
        dt2 = statistics.median(regs2)
        pipelined = {'value': R * a.steps / dt2, 'unit': 'rollouts/s', 'ms_per_step': 1e3 * dt2 / a.steps, 'timed_regions': len(regs2),
                     'what': 'the headline steps alternated over two engine contexts on two HIP streams (same kernels, same results; '
                             'the transition stages of one step overlap the decoder of the other)'}
        del m2
    # OPT-IN EXPERIMENTS, never the headline (narrower operands than the reference's fp32 arithmetic): the same steps with the decoder's
    # Linear(256, 16384) and its three large transposed convolutions on the 16-bit matrix pipe, both operands split (csrc/bf16x3.hip):
    #   mfma_bf16x3: three bf16 planes, 6 products per fp32-equivalent MAC;   mfma_f16x2: two fp16 planes, 3 products, weights scaled by 2^k
    # fp32 accumulation; every decoder-bearing fixture passes at the UNCHANGED tolerances in both modes (tests/test_gpu_parity.py::test_split_operands_*)
    PEAK_16BIT_TF = 2516.6                                         # dense bf16 / fp16 MFMA, MI355X_MICROARCH.md
    SPLIT_MODES = {'mfma_bf16x3': ('rollout_bf16x3', 6, 'bf16 x 3 planes per operand, 6 products'),
                   'mfma_f16x2': ('rollout_f16x2', 3, 'fp16 x 2 planes per operand, 3 products, weights x 2^k')}
    splits = {}
    if not a.no_extras and solo:
        n_img = 3 * S * D * R                                      # decoder rows of one step (19 200 at the default size)
        fp32k = locals().get('kern') or {}
        for opt, (xkey, nprod, dtxt) in SPLIT_MODES.items():
            peak = PEAK_16BIT_TF / nprod                            # fp32-equivalent TFLOP/s: every MAC costs `nprod` 16-bit products
            model.set_option(opt, 1)
            try:
                for k in range(4):
                    step(kk); kk += 1
                with ClockSampler(device) as clk3:
                    regs3, _, kk = timed_regions(step, a.steps, kk, rk, min_total_s=2.0, max_regions=12)
                dt3 = statistics.median(regs3)
                cls3 = {}
                for c_ in ('dec_dense_16384', 'dec_a_convT1_convT2', 'dec_b_convT3_final_reduce'):       # one class per pass (event pairs around every launch inflate a step)
                    model.prof_enable(True, classes=[c_])
                    for _ in range(5):
                        step(kk); kk += 1
                    ms_, n_ = model.prof_read()[c_]
                    cls3[c_] = ms_ / max(n_, 1)
                model.prof_enable(False)

                def kline(name, cls, macs_row):
                    ms_ = cls3[cls]
                    tf = 2.0 * macs_row * n_img / (ms_ * 1e-3) / 1e12 if ms_ > 0 else None
                    return {'name': name, 'avg_launch_ms': ms_, 'fp32_kernel_ms': (fp32k.get(cls) or {}).get('ms'), 'fp32_equivalent_tflops': tf,
                            'frac_of_16bit_peak_over_products': tf / peak if tf else None}
                ks_ = [kline('k_fc4_b3', 'dec_dense_16384', 256 * 16384), kline('k_dec_a_b3', 'dec_a_convT1_convT2', 2 * 256 * 9 * 64 * 64),
                       kline('k_dec_b_b3 (ConvT3 on the 16-bit pipe; the 32 -> 1 layer, sigmoid and sums stay fp32)', 'dec_b_convT3_final_reduce', MAC_DECB_ROW)]
                dom3 = max(ks_, key=lambda e: e['avg_launch_ms'])   # the roofline entry is the experiment's own dominant kernel
                x = {'value': R * a.steps / dt3, 'unit': 'rollouts/s', 'ms_per_step': 1e3 * dt3 / a.steps, 'timed_regions': len(regs3),
                     'dtype': dtxt + ', fp32 accumulate (Linear(256, 16384), ConvT(64,64,s1), ConvT(64,64,s2), ConvT(64,32,s2); every other kernel f32)',
                     'what': f'EXPERIMENT, not the headline: engine option {opt} = 1 on the headline workload',
                     'kernels': ks_,
                     'roofline': {'bound': 'mfma', 'kernel': dom3['name'], 'peak': peak,
                                  'unit': f'TFLOP/s (fp32-equivalent: 16-bit dense peak {PEAK_16BIT_TF} / {nprod} products)',
                                  'achieved': dom3['fp32_equivalent_tflops'], 'frac': dom3['frac_of_16bit_peak_over_products'],
                                  'avg_launch_ms': dom3['avg_launch_ms'], 'traffic': None},
                     'speedup_vs_headline': (R * a.steps / dt3) / value if rank == 0 else None}
                x['roofline'].update(clk3.report(dom3['frac_of_16bit_peak_over_products']))
                if n_img == 19200:                       # the committed PMC profile of this mode is a 19 200-image launch
                    bpl, src = committed_split_traffic(dom3['name'].split(' ')[0], opt)
                    if bpl is not None:
                        x['roofline']['traffic'] = bpl
                        x['roofline']['traffic_unit'] = 'bytes per launch'
                        x['roofline']['traffic_from_profile'] = {'file': 'profiles/' + src, 'how': 'rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE in separate passes, mean per dispatch'}
                splits[opt] = x
            except Exception as ex:          # an experiment must never cost the line its headline
                splits[opt] = {'error': repr(ex)[:300]}
            finally:
                model.set_option(opt, 0)
    if not a.no_extras:
        # every rank runs the extras (they are collective at N > 1); rank 0 attaches them
        key = 'mcts_cfg3' if world == 1 else 'mcts_cfg4_sharded'
        mc = bench_mcts(a, model, device, rk, 3, 1, with_cpu and world == 1)
        mc05 = bench_mcts(a, model, device, rk, 3, 2, False, threshold=0.5, min_total_s=2.5)
        # the same batch with every iteration run again RIGHT BEHIND it: the board slows by 3 - 5 % over the first minute of planner legs
        # (clock / temperature), so the early-stop line is compared with an adjacent full-work measurement, not with the leg run a minute earlier
        a_np = argparse.Namespace(**dict(vars(a), no_prof=True))
        mcadj = bench_mcts(a_np, model, device, rk, 3, 1, False, threshold=2.0, min_total_s=2.0)
        mc05['full_work_adjacent'] = {'value': mcadj['value'], 'ms_per_step': mcadj['ms_per_step'], 'timed_regions': mcadj['timed_regions']}
        mc05['speedup_vs_adjacent_full_work'] = mc05['value'] / mcadj['value']
        for opt, x in splits.items():
            if 'error' in x:
                continue
            # the planner with the experiment on (its expansions are 7 680-image launches: the split kernels serve them; the 960-image simulations too)
            model.set_option(opt, 1)
            mcx = None
            try:
                mcx = bench_mcts(a_np, model, device, rk, 3, 1, False, threshold=2.0, min_total_s=2.0)
            except Exception as ex:
                x['mcts_cfg3'] = {'error': repr(ex)[:300]}
            finally:
                model.set_option(opt, 0)
            if mcx is not None:
                x['mcts_cfg3'] = {'value': mcx['value'], 'unit': mcx['unit'], 'ms_per_step': mcx['ms_per_step'], 'timed_regions': mcx['timed_regions'],
                                  'speedup_vs_fp32_adjacent': mcx['value'] / mcadj['value'],
                                  'what': f'EXPERIMENT: configs[2] (64 episodes in lock-step) with {opt} = 1, against the full-work fp32 measurement taken right before the experiments'}
        single = bench_single_episode(model, device, a.samples) if world == 1 else None
        del model
        torch.cuda.empty_cache()
        ai = bench_generic(a, device, rk, 2, 1, with_cpu and world == 1)
        if rank == 0:
            out['extras'] = {key: mc, key + '_threshold_0.5': mc05, 'animalai_cfg5': ai}
            if pipelined:
                out['extras']['rollout_two_streams'] = pipelined
            for opt, x in splits.items():
                out['extras'][SPLIT_MODES[opt][0]] = x
            if single:
                out['extras']['single_episode'] = single
    emit(out)


if __name__ == '__main__':
    main()
