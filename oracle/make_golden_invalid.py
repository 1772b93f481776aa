"""ORACLE tooling -- runs ONLY in the build container (needs /root/reference).

The reference's bare-except fallback of `mcts_step_simulate` (/root/reference/src/torchmodel.py:362-367, 378-381: a habit posterior
that torch.multinomial rejects -> action 0 on that step, and at step 0 the returned Qpi is that one-hot), captured from the reference
itself with a habit network whose output bias makes the posterior invalid:

  simulate_invalid_nan   top.qpi_net.4.bias[1] = NaN   -> every logit row has a NaN, softmax is NaN everywhere
  simulate_invalid_inf   top.qpi_net.4.bias[2] = +inf  -> softmax = [0, 0, NaN, 0] (inf - inf)

Shim and noise injection are those of oracle/make_golden.py (imported, not repeated); the patched multinomial raises on invalid
probabilities exactly where torch's does.  Episode 3, depth 4, stage 60: noise rows as the lock-step planner uses them.
Fixtures hold tensors only.     Usage:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_invalid
"""
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import philox as PX
from oracle import synth
from oracle.efe_oracle import OracleModel, PhiloxNoise
from oracle.make_golden import load_reference, GOLD, npy

CASES = {'simulate_invalid_nan': (1, float('nan')), 'simulate_invalid_inf': (2, float('inf'))}
WSEED, GAIN, NSEED, DEPTH, STAGE, EPISODE = 1234, 1.15, 7, 4, 60, 3


def poisoned_weights(case):
    idx, val = CASES[case]
    w = {k: np.array(v, copy=True) for k, v in synth.make_weights(WSEED, GAIN).items()}
    w['top.qpi_net.4.bias'][idx] = val
    return w


def main():
    torch.set_grad_enabled(False)
    man_path = os.path.join(GOLD, 'MANIFEST.json')
    manifest = json.load(open(man_path))
    for case in CASES:
        weights = poisoned_weights(case)
        model, inj, ref_mcts, state = load_reference(weights, NSEED)
        start = torch.from_numpy(PX.uniform_fill(4, (10,), 81, -1.0, 1.0))
        inj.stage = STAGE
        state['episode'] = EPISODE
        G, pi0, Qpi = model.mcts_step_simulate(start, DEPTH, use_means=False)
        assert not inj.q
        assert np.array_equal(npy(pi0), np.eye(4, dtype=np.float32)[[0] * DEPTH]) and np.array_equal(npy(Qpi), [1, 0, 0, 0]), (pi0, Qpi)
        orc = OracleModel(weights, PhiloxNoise(NSEED))
        oG, opi0, oQpi = orc.mcts_step_simulate(start, DEPTH, False, STAGE, episode=EPISODE)
        assert abs(G - oG) <= 1e-3 and np.array_equal(npy(pi0), npy(opi0)) and np.array_equal(npy(Qpi), npy(oQpi)), (G, oG)
        arrs = dict(start=start, depth=DEPTH, stage=STAGE, episode=EPISODE, G=G, pi0=pi0, Qpi=Qpi, wseed=WSEED, gain=GAIN, nseed=NSEED,
                    bias_index=CASES[case][0], bias_value=np.float32(CASES[case][1]))
        np.savez_compressed(os.path.join(GOLD, case + '.npz'), **{k: npy(v) for k, v in arrs.items()})
        manifest['cases'][case] = sorted(arrs.keys())
        print(case, 'G', G, 'oracle', oG)
    json.dump(manifest, open(man_path, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
