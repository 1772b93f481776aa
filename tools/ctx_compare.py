#!/usr/bin/env python3
"""dev tool: the same decoder kernels timed (HIP events) in different calling contexts on one box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daimc_amd  # noqa: E402

model = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=7)
dev = 'cuda:0'


def timed(name, fn, reps=4):
    fn(); torch.cuda.synchronize()
    model.prof_enable(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    r = model.prof_read()
    print(f'{name:34s} ' + '  '.join(f'{k} {ms / reps:.3f}' for k, (ms, n) in r.items() if n and ms / reps > 0.3))


s = torch.randn(19200, 10, device=dev)
timed('decoder(randn)', lambda: model.model_down.decoder(s))
timed('decoder(3*randn)', lambda: model.model_down.decoder(3 * s))
timed('decoder(0.1*randn)', lambda: model.model_down.decoder(0.1 * s))
s0 = torch.randn(640, 10, device=dev); pi0 = torch.eye(4, device=dev).repeat(160, 1)
timed('calculate_G(M=640, S=10)', lambda: model.calculate_G(s0, pi0, 10))
o = torch.rand(128, 1, 64, 64, device=dev); pi = torch.eye(4, device=dev).repeat(32, 1)
timed('calculate_G_repeated(128, D=5)', lambda: model.calculate_G_repeated(o, pi, 5, False, 10))
timed('calculate_G_repeated(640, D=1)', lambda: model.calculate_G_repeated(torch.rand(640, 1, 64, 64, device=dev), torch.eye(4, device=dev).repeat(160, 1), 1, False, 10))
