#!/usr/bin/env python3
"""dev tool: latency of one efe_simulate call (E episodes, depth 5) and of a one-episode planner iteration with the simulation chain on one
workgroup (sim_split = 0) and on eight (sim_split = 1)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daimc_amd
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=7)
for E in (1, 8, 16):
    s = torch.randn(E, 10, device='cuda:0')
    for split in (0, 1):
        m.set_option('sim_split', split)
        for _ in range(5): m.simulate_batch(s, 5)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(200): m.simulate_batch(s, 5)
        torch.cuda.synchronize()
        print(f'E={E} sim_split={split}: {(time.perf_counter() - t) / 200 * 1e3:.3f} ms per simulate_batch', flush=True)
frame = torch.rand(1, 1, 64, 64, device='cuda:0')
for split in (0, 1):
    m.set_option('sim_split', split)
    q = daimc_amd.MCTS_Params(); q.repeats, q.simulation_depth, q.threshold, q.use_means, q.samples = 50, 5, 2.0, False, 10
    for _ in range(2): daimc_amd.active_inference_mcts_batch(m, frame, q, o_shape=(1, 64, 64))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): daimc_amd.active_inference_mcts_batch(m, frame, q, o_shape=(1, 64, 64))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f'one-episode decision sim_split={split}: {dt * 1e3:.2f} ms = {dt / 51 * 1e3:.3f} ms per iteration', flush=True)
