"""Development probe: run the generic-geometry decoder / encoder / calculate_G on scratch memory pre-filled with a poison byte
(engine option `poison`) and compare with the unpoisoned result: a kernel that reads scratch it never wrote shows up as a difference
(or as a GPU fault).  usage: python tools/poison_probe.py A C R [byte]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import daimc_amd
from oracle import synth, philox as PX

A, C, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
byte = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0xff
w = synth.make_weights(77 + R, 1.15, A, C, R) if (C, R) != (1, 64) else synth.make_weights(1234, 1.15)
m = daimc_amd.ActiveInferenceModel(10, A, 0.0, 1.0, 1.0, colour_channels=C, resolution=R, device='cuda:0', seed=5, init_weights=False)
m.load_flat_weights(w)
M = 3
s = PX.uniform_fill(4, (M, 10), 60, -1.5, 1.5)
pi = np.eye(A, dtype=np.float32)[np.arange(M) % A]
fr = synth.make_frames_rgb(12, M, C, R) if (C, R) != (1, 64) else synth.make_frames(12, M)[:, 0][:, None]


def run():
    po = m.model_down.decoder(s, stage=2, pass_=PX.PASS_D1)
    es = m.model_down.encoder_with_sample(fr, stage=2, pass_=PX.PASS_E1)
    G = m.calculate_G(s, pi, samples=2, stage=2)
    R_ = m.calculate_G_repeated(fr, pi, steps=2, calc_mean=False, samples=2, stage=3)
    torch.cuda.synchronize()
    return [po.cpu(), es[1].cpu(), G[0].cpu(), G[1][0].cpu(), R_[0].cpu()]


ref = run()
m.set_option('poison', byte)
for k in range(2):
    out = run()
    for i, (a, b) in enumerate(zip(ref, out)):
        same = torch.equal(a, b)
        print(f'pass {k} output {i}: {"identical" if same else "DIFFERENT max|d| = %g" % float((a - b).abs().max())}')
print('done', A, C, R, hex(byte))
