#!/usr/bin/env python3
"""dev tool: one single-episode planner decision (lock-step planner at E = 1, S = 10, graph replay) for a rocprofv3 kernel trace"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daimc_amd
from bench import synth_frames
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=1)
fr = synth_frames(1, torch.device('cuda:0'), seed=300)
q = daimc_amd.MCTS_Params(); q.repeats, q.simulation_depth, q.threshold, q.use_means, q.samples = 50, 5, 2.0, False, 10
q.use_graph = len(sys.argv) > 1 and sys.argv[1] == 'graph'
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    daimc_amd.active_inference_mcts_batch(m, fr, q, o_shape=(1, 64, 64))
    torch.cuda.synchronize(); print('decision ms', 1e3 * (time.perf_counter() - t0))
