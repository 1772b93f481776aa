"""timing breakdown of one lock-step MCTS iteration (dev tool)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daimc_amd
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=1)
E = 64
s = torch.randn(E * 4, 10, device='cuda') * 0.3
pi = torch.eye(4, device='cuda').repeat(E, 1)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print('calculate_G 256 rows S=10 : %.2f ms' % t(lambda: m.calculate_G(s, pi, samples=10)))
print('calculate_G 256 rows S=1  : %.2f ms' % t(lambda: m.calculate_G(s, pi, samples=1)))
print('calculate_G_mean 256 rows : %.2f ms' % t(lambda: m.calculate_G_mean(s, pi)))
print('simulate_batch 64 x 5     : %.2f ms' % t(lambda: m.simulate_batch(s[:E], 5)))
print('simulate_batch 64 x 3     : %.2f ms' % t(lambda: m.simulate_batch(s[:E], 3)))
print('encoder 64                : %.2f ms' % t(lambda: m.model_down.encoder(torch.zeros(E, 1, 64, 64, device='cuda'))))
print('habit 64                  : %.2f ms' % t(lambda: m.model_top.encode_s(s[:E])))
print('rollout 128 D5 S10        : %.2f ms' % t(lambda: m.calculate_G_repeated(torch.zeros(128, 1, 64, 64, device='cuda'), pi[:128], steps=5, samples=10)))
m.prof_enable(True)
m.calculate_G(s, pi, samples=10); print({k: (round(v[0], 3), v[1]) for k, v in m.prof_read().items() if v[1]})
m.simulate_batch(s[:E], 5); print({k: (round(v[0], 3), v[1]) for k, v in m.prof_read().items() if v[1]})
