#!/usr/bin/env python3
"""dev tool: where does the generic decoder's image differ between two engine builds (EFE_LIB_PATH) / from the oracle?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daimc_amd
from oracle import philox as PX, synth, efe_oracle as EO
A, C, R = 3, int(os.environ.get('DBG_C', 3)), int(os.environ.get('DBG_R', 84))
w = synth.make_weights(4321, 1.15, A, C, R)
m = daimc_amd.ActiveInferenceModel(10, A, 0.0, 1.0, 1.0, colour_channels=C, resolution=R, device='cuda:0', seed=9, init_weights=False)
m.load_flat_weights(w)
orc = EO.OracleModel(w, EO.PhiloxNoise(9), pi_dim=A, channels=C, resolution=R)
M = 2
s = PX.uniform_fill(3, (M, 10), 50, -1.5, 1.5)
with torch.no_grad():
    opo = orc.decoder(torch.from_numpy(s), PX.PASS_D1, 0, 5).numpy()
po = m.model_down.decoder(s, stage=5, pass_=PX.PASS_D1).cpu().numpy()
d = np.abs(po - opo) > 1e-5
print('mismatch', d.sum(), 'of', d.size)
for i in range(M):
    for c in range(C):
        ys, xs = np.nonzero(d[i, c])
        print('img', i, 'ch', c, 'n', len(ys), 'rows', sorted(set(ys.tolist()))[:40], 'cols', sorted(set(xs.tolist()))[:60])
