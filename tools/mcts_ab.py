"""dev tool: lock-step MCTS iteration time with / without the simulate overlap"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daimc_amd
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=1)
E = 64
frames = torch.rand(E, 1, 64, 64, device='cuda')
for ov in (True, False, True, False):
    p = daimc_amd.MCTS_Params(); p.repeats, p.simulation_depth, p.use_means, p.threshold, p.samples = 50, 5, False, 2.0, 10
    p.overlap_simulate = ov
    daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2):
        daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
    torch.cuda.synchronize()
    print('overlap', ov, 'ms per iteration %.3f' % ((time.perf_counter() - t0) / 2 / 51 * 1e3))
