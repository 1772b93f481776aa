"""single-episode latency of the drop-in API (the reference's own usage pattern: batch 4 / batch 1 calls)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daimc_amd
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=1)
s4 = torch.randn(4, 10, device='cuda') * 0.3
frame = torch.rand(64, 64, 1, device='cuda')
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('calculate_G_mean (4 rows)        %.3f ms' % t(lambda: m.calculate_G_mean(s4, m.pi_one_hot)))
print('calculate_G (4 rows, S=10)       %.3f ms' % t(lambda: m.calculate_G(s4, m.pi_one_hot, samples=10)))
print('mcts_step_simulate (depth 3)     %.3f ms' % t(lambda: m.mcts_step_simulate(s4[0], 3)))
print('calculate_G_4_repeated D5 S10    %.3f ms' % t(lambda: m.calculate_G_4_repeated(frame.reshape(1, 1, 64, 64).repeat(4, 1, 1, 1), steps=5, samples=10)))
p = daimc_amd.MCTS_Params(); p.repeats = 50; p.simulation_depth = 5; p.threshold = 2.0
print('active_inference_mcts 50 repeats %.1f ms' % t(lambda: daimc_amd.active_inference_mcts(m, frame, p, o_shape=(1, 64, 64)), n=3))
p.use_means = False
print('  ... use_means=False            %.1f ms' % t(lambda: daimc_amd.active_inference_mcts(m, frame, p, o_shape=(1, 64, 64)), n=3))
# lock-step planner with one episode: every iteration launched vs the iteration replayed from a captured hipGraph (device noise)
fr1 = frame.reshape(1, 1, 64, 64)
for samples in (1, 10):
    for mode in (False, True):
        q = daimc_amd.MCTS_Params(); q.repeats = 50; q.simulation_depth = 5; q.threshold = 2.0; q.use_means = False; q.samples = samples; q.use_graph = mode
        print('active_inference_mcts_batch E=1, 50 repeats, samples=%d, %s  %.1f ms' % (samples, 'hipGraph replay ' if mode else 'launched        ',
              t(lambda: daimc_amd.active_inference_mcts_batch(m, fr1, q, o_shape=(1, 64, 64)), n=5)))
fr4 = torch.rand(4, 1, 64, 64, device='cuda')
for mode in (False, True):
    q = daimc_amd.MCTS_Params(); q.repeats = 50; q.simulation_depth = 5; q.threshold = 2.0; q.use_means = False; q.samples = 10; q.use_graph = mode
    print('active_inference_mcts_batch E=4, 50 repeats, samples=10, %s  %.1f ms' % ('hipGraph replay ' if mode else 'launched        ',
          t(lambda: daimc_amd.active_inference_mcts_batch(m, fr4, q, o_shape=(1, 64, 64)), n=5)))
fr64 = torch.rand(64, 1, 64, 64, device='cuda')
for mode in (False, True):
    q = daimc_amd.MCTS_Params(); q.repeats = 50; q.simulation_depth = 5; q.threshold = 2.0; q.use_means = False; q.samples = 10; q.use_graph = mode
    print('active_inference_mcts_batch E=64, 50 repeats, samples=10, %s  %.1f ms' % ('hipGraph replay ' if mode else 'launched        ',
          t(lambda: daimc_amd.active_inference_mcts_batch(m, fr64, q, o_shape=(1, 64, 64)), n=3)))
