"""dev tool: what the per-iteration stop check costs a ONE-episode decision: threshold out of reach (no check), a threshold that can trigger
(0.7499 < 1 - 1/4) with the immediate read of the active count (lagged_single = False) and with the lagged snapshot (the default)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import daimc_amd
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=7)
frame = torch.rand(1, 1, 64, 64, device='cuda:0')
for thr, lag in ((2.0, None), (0.7499, 0), (0.7499, 1)):
    q = daimc_amd.MCTS_Params(); q.repeats, q.simulation_depth, q.threshold, q.use_means, q.samples = 50, 5, thr, False, 10
    if lag is not None: q.lagged_single = bool(lag)
    for _ in range(5): daimc_amd.active_inference_mcts_batch(m, frame, q, o_shape=(1, 64, 64))
    torch.cuda.synchronize(); best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(10): out = daimc_amd.active_inference_mcts_batch(m, frame, q, o_shape=(1, 64, 64))
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10 / 50 * 1e3)
    print(thr, lag, f'{best:.4f} ms per iteration', out[0][0][1])
