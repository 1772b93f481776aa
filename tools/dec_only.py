#!/usr/bin/env python3
"""dev tool: time the decoder kernel classes on a stand-alone decoder call (N rows) with HIP events, to compare with the
isolated micro-benchmarks (tools/ubench/dec_?_bench.hip) and with the kernels inside a full rollout step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daimc_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 19200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
model = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=7)
s = torch.randn(N, 10, device='cuda:0')
model.model_down.decoder(s)
torch.cuda.synchronize()
for blk in range(reps):
    model.prof_enable(True)
    for _ in range(6):
        model.model_down.decoder(s)
    torch.cuda.synchronize()
    r = model.prof_read()
    print(f'block {blk}: ' + '  '.join(f'{k} {ms / 6:.3f}' for k, (ms, n) in r.items() if n and ms / 6 > 0.5))
