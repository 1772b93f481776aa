#!/usr/bin/env python3
"""dev tool: what does a row mask buy?  One 256-row x 10-sample expansion (calculate_G) and one 64-episode depth-5 simulation with
0 / 25 / 50 / 100 % of the episodes dead; HIP-event time per call."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daimc_amd
from daimc_amd.model import Rows
E, A = 64, 4
m = daimc_amd.ActiveInferenceModel(10, A, 0.0, 1.0, 1.0, device='cuda:0', seed=7)
s = torch.randn(E * A, 10, device='cuda:0'); pi = torch.eye(A, device='cuda:0').repeat(E, 1)
ls = torch.randn(E, 10, device='cuda:0')
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for dead in (0.0, 0.25, 0.5, 1.0):
    mask = torch.ones(E, dtype=torch.uint8, device='cuda:0'); mask[:int(E * dead)] = 0
    ra, r1 = Rows(mask=mask, rows_per_entry=A), Rows(mask=mask)
    te = t(lambda: m.calculate_G(s, pi, samples=10, rows=ra))
    ts = t(lambda: m.simulate_batch(ls, 5, use_means=False, rows=r1))
    m.prof_enable(True); m.calculate_G(s, pi, samples=10, rows=ra); torch.cuda.synchronize(); r = m.prof_read(); m.prof_enable(False)
    print(f'dead {dead:.2f}: expansion {te:.3f} ms  simulation {ts:.3f} ms  ', '  '.join(f'{k} {ms:.3f}' for k, (ms, n) in r.items() if n))
