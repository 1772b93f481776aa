"""N > 1 ranks of the HIP ENGINE (-m gpu): two processes, one engine context each, shard the episodes with global row /
episode offsets, run rollouts and the lock-step planner on their shard, gather the action posteriors / root visit
distributions, and must reproduce (a) the single-process result BIT FOR BIT and (b) the per-episode fixtures captured from
the reference planner with Node.expand(samples=10) (BASELINE configs[2]/[3] shape).

With >= 2 visible GPUs the ranks use cuda:0 / cuda:1 and the RCCL ("nccl") backend; on a 1-GPU box both ranks share cuda:0
(two processes, two contexts) and the gather runs over gloo -- the engine-side code path (efe_create per process, row
offsets, gather of engine-produced tensors) is the same."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
N_EP = 8


def _params(g):
    import daimc_amd
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.use_means, p.threshold = int(g['repeats']), int(g['simulation_depth']), False, float(g['threshold'])
    p.samples = int(g['samples'])
    return p


def _run_shard(device, start, count, g):
    """rollout posteriors + lock-step plans of episodes [start, start+count) with injected Philox normals"""
    import daimc_amd
    from oracle import synth, philox as PX
    m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device=device, seed=int(g['nseed']), init_weights=False)
    m.load_flat_weights(synth.make_weights(int(g['wseed']), float(g['gain'])))
    m.eps_source, m.u_source = PX.normals, PX.uniforms
    frames = torch.from_numpy(g['frames'][start:start + count])
    o = frames.repeat_interleave(4, dim=0)
    pi = torch.eye(4).repeat(count, 1)
    G, _, _ = m.calculate_G_repeated(o, pi, steps=2, samples=3, stage=0, row_offset=4 * start)
    P, _ = m.action_posterior(G)
    m._stage = int(g['stage'])
    out, visits = daimc_amd.active_inference_mcts_batch(m, frames, _params(g), o_shape=(1, 64, 64), episode_offset=start)
    return P, visits.to(P.device), G, out


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist
    torch.set_num_threads(2)
    multi = torch.cuda.device_count() >= world
    dev = torch.device('cuda', rank if multi else 0)
    torch.cuda.set_device(dev)
    if multi:
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    import daimc_amd
    g = load_golden('mcts_batch_s10')
    start, count = daimc_amd.episode_shard(N_EP, world, rank)
    P, visits, G, out = _run_shard(dev, start, count, g)
    if not multi:                                   # gloo: gather host copies of the engine's outputs
        P, visits = P.cpu(), visits.cpu()
    allP = daimc_amd.gather_action_posteriors(P, N_EP)
    allV = daimc_amd.gather_action_posteriors(visits, N_EP)
    assert allP.shape == (N_EP, 4) and allV.shape == (N_EP, 4)
    torch.save({'out': out, 'G': G.cpu()}, os.path.join(out_dir, f'rank{rank}.pt'))
    if rank == 0:
        torch.save({'P': allP.cpu(), 'V': allV.cpu(), 'backend': dist.get_backend()}, os.path.join(out_dir, 'gathered.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_engine_equals_single_process_and_reference(tmp_path):
    import torch.multiprocessing as mp
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / 'gathered.pt')
    # with two or more GPUs the gather must have gone through RCCL (one device per rank), never the gloo fallback
    assert got['backend'] == ('nccl' if torch.cuda.device_count() >= 2 else 'gloo'), got['backend']
    g = load_golden('mcts_batch_s10')
    P1, V1, G1, out1 = _run_shard(torch.device('cuda', 0), 0, N_EP, g)
    # (a) sharded == unsharded, bit for bit (noise keyed by global rows; no cross-row arithmetic anywhere)
    assert torch.equal(got['P'], P1.cpu()), got['backend']
    assert torch.equal(got['V'], V1.cpu())
    outs = []
    Gs = []
    for r in range(2):
        d = torch.load(tmp_path / f'rank{r}.pt')
        outs += d['out']; Gs.append(d['G'])
    assert torch.equal(torch.cat(Gs), G1.cpu())
    assert [o[0] for o in outs] == [o[0] for o in out1] and [o[3] for o in outs] == [o[3] for o in out1]
    assert [o[4] for o in outs] == [o[4] for o in out1]
    # (b) every episode == the reference planner run on that episode alone (fixture captured with Node.expand(samples=10))
    for e in range(N_EP):
        path, reps, explored, all_paths, all_G = outs[e]
        n = int(g['n_paths'][e])
        assert reps == int(g['repeats_done'][e]) and explored == int(g['states_explored'][e]) and len(all_paths) == n
        assert all_paths == [[int(a) for a in row if a >= 0] for row in g['all_paths'][e][:n]]
        np.testing.assert_allclose(np.array(all_G), g['all_paths_G'][e][:n], atol=5e-6 * 2800 + 1e-3)
        assert [int(a) for a in path] == [int(a) for a in g['final_path'][e] if a >= 0]
        np.testing.assert_array_equal(got['V'][e].numpy(), g['root_N'][e] / g['root_N'][e].sum())


def _bench_line(args, timeout):
    import json
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                        # ONE JSON line on stdout, whatever the rank count
    return json.loads(lines[0])


def test_bench_rccl_path_at_one_rank():
    """the N > 1 code path of bench.py through RCCL itself on a 1-GPU box: `--force-dist` initialises the "nccl" process group with one
    rank, so the device-tensor all_gather of the action posteriors, the max-over-ranks timing and the collective fields of the line run
    through the library the 8-GPU line will use"""
    d = _bench_line(['--force-dist', '--no-extras', '--steps', '3', '--warmup', '2', '--min-seconds', '0.5'], 600)
    assert d['n_gpus'] == 1 and d['rccl_ranks'] == 1 and d['backend'] == 'nccl'
    assert len(d['per_rank_ms_per_step']) == 1 and d['all_gather_ms'] > 0
    assert d['roofline']['frac'] > 0.3 and d['roofline']['bound'] == 'mfma' and 'sclk_mhz_under_load' in d['roofline']
    assert d['cpu_baseline']['value'] > 0 and d['cpu_baseline']['kind'] == 'port'
    assert d['value'] > 1000 and d['config']['rows_per_gpu'] == 128


@pytest.mark.parametrize('n', [4, 8])
def test_bench_launcher_at_n_ranks_on_one_device(n):
    """`python bench.py --gpus N --share-device` at 4 and at the 8 ranks of the driver's scaling run: the self-launcher
    (torch.distributed.run on 127.0.0.1), episode sharding with global row offsets, region-count agreement across ranks, rank 0's single
    JSON line with the per-rank figures, the per-rank (device, PCI bus id) list and the CPU leg -- every rank on cuda:0 over gloo (a
    rehearsal of the 8-GPU line, which needs a multi-GPU node).  Without --share-device the same command REFUSES to put two ranks on one
    GPU (bench.py Ranks.check_devices: efe_get_device of every rank's engine context, gathered)."""
    d = _bench_line(['--gpus', str(n), '--share-device', '--no-extras', '--no-prof', '--steps', '2', '--warmup', '1', '--min-seconds', '0.5'], 1200)
    assert d['n_gpus'] == n and d['ranks'] == n and 'rccl_ranks' not in d and d['backend'] == 'gloo' and d.get('share_device') is True
    assert len(d['per_rank_ms_per_step']) == n and d['all_gather_ms'] > 0 and d['scaling'] == 'weak'
    assert len(d['devices']) == n and all(dev[0] == 0 and dev[1] == d['devices'][0][1] and len(dev[1]) >= 7 for dev in d['devices'])
    assert d['cpu_baseline']['value'] > 0
    assert d['config']['rows_per_gpu'] == 128 and d['value'] > 1000


def test_engine_context_reports_its_device():
    """efe_get_device: the context's HIP device index and PCI bus id -- what bench.py's rank -> GPU check reads"""
    import daimc_amd
    import torch
    m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=1)
    idx, bus = m.engine_device()
    assert idx == 0 and bus.count(':') == 2 and '.' in bus
