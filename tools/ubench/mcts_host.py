"""dev tool: where the host spends its time in one 64-episode decision batch (cProfile), and how long the GPU has nothing queued
at both ends of the batch."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import daimc_amd
from daimc_amd import mcts as M
E = 64
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=3)
frames = np.random.rand(E, 64, 64, 1).astype(np.float32)
p = M.MCTS_Params(); p.repeats = 50; p.threshold = 2.0; p.simulation_depth = 5; p.samples = 10; p.use_means = False
for _ in range(2): M.active_inference_mcts_batch(m, frames, p)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter()
for _ in range(3): M.active_inference_mcts_batch(m, frames, p)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 3
pr.disable()
print('ms per decision batch', dt * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18); print(s.getvalue()[:3500])
