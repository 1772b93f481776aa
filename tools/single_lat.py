#!/usr/bin/env python3
"""dev tool: single-episode decision latency (lock-step planner at E = 1, S = 10), simulation on the second stream or not, launched / graph"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daimc_amd
from bench import synth_frames
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=1)
for E in (1, 4):
    fr = synth_frames(E, torch.device('cuda:0'), seed=300)
    for ov in (False, True):
        for gr in (False, True):
            q = daimc_amd.MCTS_Params(); q.repeats, q.simulation_depth, q.threshold, q.use_means, q.samples = 50, 5, 2.0, False, 10
            q.use_graph, q.overlap_simulate = gr, ov
            ts = []
            for _ in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                daimc_amd.active_inference_mcts_batch(m, fr, q, o_shape=(1, 64, 64))
                torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
            print(f'E={E} overlap={ov} graph={gr}: decision {min(ts[1:]):.1f} ms = {min(ts[1:]) / 51:.3f} ms per iteration', flush=True)
