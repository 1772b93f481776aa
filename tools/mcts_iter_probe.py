#!/usr/bin/env python3
"""dev tool: GPU time of every planner iteration against the number of live episodes (threshold 0.5, 64 episodes)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daimc_amd
from daimc_amd import mcts as M
from bench import synth_frames
dev = torch.device('cuda:0')
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=1)
frames = synth_frames(64, dev, seed=200)
p = daimc_amd.MCTS_Params(); p.repeats, p.simulation_depth, p.use_means, p.threshold, p.samples = 50, 5, False, 0.5, 10
for kw in ({'threshold': 2.0}, {'threshold': 0.5}, {'threshold': 0.5, 'check_every': 16}):
    for k, v in kw.items(): setattr(p, k, v)
    daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
    # instrument: an event after every efe_mcts_step call
    evs, lives = [], []
    orig = M.BatchedMCTS._call
    def call(self, fn, *a):
        r = orig(self, fn, *a)
        if getattr(fn, '__name__', '') == 'efe_mcts_step':
            e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e); lives.append(self.active.clone())
        return r
    M.BatchedMCTS._call = call
    torch.cuda.synchronize(); e_in = torch.cuda.Event(enable_timing=True); e_in.record(); t0 = time.perf_counter()
    out, _ = daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
    t1 = time.perf_counter(); e_out = torch.cuda.Event(enable_timing=True); e_out.record()
    M.BatchedMCTS._call = orig
    torch.cuda.synchronize()
    print('  wall %.1f ms; GPU: entry -> first step %.2f ms, steps %.1f ms, last step -> exit %.2f ms' % (1e3 * (t1 - t0), e_in.elapsed_time(evs[0]), evs[0].elapsed_time(evs[-1]), evs[-1].elapsed_time(e_out)))
    ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]
    nl = [int(l.sum()) for l in lives]
    print(kw, 'iterations', len(evs), 'total', round(sum(ts), 1), 'ms')
    print('  live :', nl[::3])
    print('  ms   :', [round(t, 2) for t in ts[::3]])
