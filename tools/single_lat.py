"""dev tool: wall time of one-episode planner decisions (lock-step planner at E = 1, S = 10, depth 5, 50 iterations), ms per iteration.
EFE_ENGINE_OPTS passes through (A/B of engine options)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daimc_amd
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=7)
frame = torch.rand(1, 1, 64, 64, device='cuda:0')
q = daimc_amd.MCTS_Params(); q.repeats, q.simulation_depth, q.threshold, q.use_means, q.samples = 50, 5, 2.0, False, 10
for _ in range(5):
    daimc_amd.active_inference_mcts_batch(m, frame, q, o_shape=(1, 64, 64))
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(10):
        daimc_amd.active_inference_mcts_batch(m, frame, q, o_shape=(1, 64, 64))
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 10 / 50 * 1e3)
print(f'{os.environ.get("EFE_ENGINE_OPTS", "")}: {best:.4f} ms per iteration')
