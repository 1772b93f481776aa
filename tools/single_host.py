"""dev tool: host-side cost of a ONE-episode decision (lock-step planner at E = 1, S = 10, depth 5, 50 iterations): wall time of the enqueue
loop alone (host returns before the GPU is done) against the synchronised time, and a cProfile of where the host spends it."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import daimc_amd
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=7)
frame = torch.rand(1, 1, 64, 64, device='cuda:0')
q = daimc_amd.MCTS_Params(); q.repeats, q.simulation_depth, q.threshold, q.use_means, q.samples = 50, 5, 2.0, False, 10
for _ in range(5): daimc_amd.active_inference_mcts_batch(m, frame, q, o_shape=(1, 64, 64))
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N): daimc_amd.active_inference_mcts_batch(m, frame, q, o_shape=(1, 64, 64))
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'per iteration: host loop {(t1 - t0) / N / 50 * 1e3:.4f} ms, synchronised {(t2 - t0) / N / 50 * 1e3:.4f} ms')
pr = cProfile.Profile(); pr.enable()
for _ in range(N): daimc_amd.active_inference_mcts_batch(m, frame, q, o_shape=(1, 64, 64))
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:4500])
