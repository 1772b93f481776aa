#!/usr/bin/env python3
"""dev tool: what does the early stop buy the lock-step planner?  64 episodes, threshold 2.0 vs 0.5 (skip_stopped on / off)."""
import os, sys, time, copy
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daimc_amd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_frames
E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda:0')
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=7)
frames = synth_frames(E, dev, seed=200)
def run(th, skip, n=3, **kw):
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.use_means, p.threshold, p.samples = 50, 5, False, th, 10
    p.skip_stopped = skip
    for k, v in kw.items(): setattr(p, k, v)
    out, _ = daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out, _ = daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    its = [o[1] for o in out]
    print(f'threshold {th} skip {skip} {kw}: {1e3 * dt:.1f} ms per batch = {E / dt:.1f} decisions/s; iterations mean {np.mean(its):.1f} min {min(its)} max {max(its)}; '
          f'work fraction {np.sum(its) / (50.0 * E):.3f}', flush=True)
run(2.0, True); run(0.5, True); run(0.5, True, compact_stopped=False); run(0.5, True, check_every=4, compact_min_dead=0.03); run(0.5, True, check_every=2, compact_min_dead=0.015); run(0.5, True, check_every=1, compact_min_dead=0.015); run(2.0, True); run(0.5, True)
