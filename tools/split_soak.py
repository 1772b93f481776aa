"""dev tool: the split-operand decoder kernels against the exact fp32 kernels over weight gains and seeds (2 000-row decoder launches: k_fc4_b3 +
k_dec_a_b3 + k_dec_b_b3): max |image difference| per mode, and calculate_G terms on 200 rows x 4 samples."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daimc_amd
from oracle import synth, philox as PX
worst = {}
for gain in (1.0, 1.35, 2.0, 3.0):
    for wseed in (1234, 77):
        m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=11, init_weights=False)
        m.load_flat_weights(synth.make_weights(wseed, gain))
        s = PX.uniform_fill(21, (2000, 10), 500 + wseed, -2.0, 2.0)
        s0 = PX.uniform_fill(22, (200, 10), 600 + wseed, -1.5, 1.5); pi0 = np.eye(4, dtype=np.float32)[np.arange(200) % 4]
        ref = m.model_down.decoder(s, stage=3, pass_=PX.PASS_D2A)
        refG = m.calculate_G(s0, pi0, samples=4, stage=8)
        for opt in ('mfma_bf16x3', 'mfma_f16x2'):
            m.set_option(opt, 1)
            got = m.model_down.decoder(s, stage=3, pass_=PX.PASS_D2A)
            gotG = m.calculate_G(s0, pi0, samples=4, stage=8)
            m.set_option(opt, 0)
            d = float((got - ref).abs().max()); dG = float((gotG[0] - refG[0]).abs().max()); scale = float(refG[1][2].abs().max())
            assert torch.isfinite(got).all() and torch.isfinite(gotG[0]).all()
            k = (opt, gain)
            worst[k] = (max(worst.get(k, (0, 0, 0))[0], d), max(worst.get(k, (0, 0, 0))[1], dG), scale)
for (opt, gain), (d, dG, sc) in sorted(worst.items()):
    print(f'{opt:12s} gain {gain:4.2f}: max |image - fp32 kernels| = {d:.3e}   max |G - fp32 kernels| = {dG:.3e}  (|term2| up to {sc:.1f})')
