import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daimc_amd
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=1)
E = 64
p = daimc_amd.MCTS_Params(); p.repeats, p.simulation_depth, p.use_means, p.threshold, p.samples = 20, 5, False, 2.0, 10
frames = torch.rand(E, 1, 64, 64, device='cuda')
daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
torch.cuda.synchronize()
pr.disable()
print('wall per iteration ms', (time.perf_counter() - t0) / 20 * 1e3)
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
