"""dev tool: BASELINE configs[1] steps alternated over two engine contexts on two HIP streams (step k+1's transition chain and small
launches run beside step k's decoder kernels) vs one context on one stream"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daimc_amd
dev = torch.device('cuda', 0)
R, D, S = 128, 5, 10
m0 = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device=dev, seed=1)
m1 = m0.replica()
for m in (m0, m1):
    m.reserve(R, D, S)
o = torch.rand(R, 1, 64, 64, device=dev); pi = torch.eye(4, device=dev).repeat(R // 4, 1)
s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
def run(n, two):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        m, st = (m0, s0) if (not two or k % 2 == 0) else (m1, s1)
        with torch.cuda.stream(st):
            G, _, _ = m.calculate_G_repeated(o, pi, steps=D, samples=S, stage=k * D)
            m.action_posterior(G)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for two in (False, True, False, True):
    run(6, two)
    ms = run(60, two)
    print('two streams' if two else 'one stream ', '%.3f ms/step -> %.1f rollouts/s' % (ms, R / ms * 1e3))
