"""ORACLE tooling -- runs ONLY in the build container (needs /root/reference).

Distribution fixtures of the reference under ITS OWN random number generator ("Device-Philox mode is validated
statistically").  Every other fixture in tests/golden/ is captured with `F.dropout` / `torch.randn_like` / `torch.multinomial`
patched to the build's Philox stream (oracle/make_golden.py), so the oracle, the injector and the engine share one keying scheme; a
keying mistake that correlated two draws the reference makes independently (/root/reference/src/torchmodel.py:274-291: the loop-1
and loop-2 masks are exactly what term2 measures) would pass all of them.  Here NOTHING of torch's noise is patched: the shimmed
reference (cv2 stub, qs_net[9] = Linear(576, 256), precision = float32 -- the three items of SURVEY appendix B) draws from torch's
global generator after `torch.manual_seed`, and the fixtures hold the SAMPLES (not summaries) of

  stats_calcG     N_G   calls of calculate_G(s0 x 4, eye(4), samples=3)        (torchmodel.py:270-300):  G, term0, term1, term2,
                  term2_1, term2_2 per row  (term2_1 / term2_2 are read off a recording wrapper around `entropy_bernoulli`,
                  torchmodel.py:289,292 -- it returns the reference's own value and draws nothing)
  stats_rollout   N_R   calls of calculate_G_repeated(o x 4, eye(4), steps=2, samples=2)  (torchmodel.py:227-245): sum_G and terms
                  (root encoder + reparameterisation, the state carried between stages)
  stats_simulate  N_S   calls of mcts_step_simulate(start, depth=5)            (torchmodel.py:354-393):  G, the sampled actions,
                  Qpi of the first step

against which the engine in device-noise mode (its production mode) and the oracle with PhiloxNoise are compared as two-sample
statistics (tests/test_noise_statistics.py).  Fixtures hold tensors only.

Usage:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_stats        (about 6 minutes on 8 cores)
"""
import json
import os
import sys
import time
import types

import numpy as np

sys.dont_write_bytecode = True
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import philox as PX
from oracle import synth

REF = '/root/reference'
GOLD = os.path.join(ROOT, 'tests', 'golden')
_ORIG = (F.dropout, torch.randn_like, torch.multinomial)       # taken before anything could patch them

WSEED, GAIN = 1234, 1.15
N_G, S_G = 4096, 3
N_R, D_R, S_R = 1024, 2, 2
N_S, DEPTH_S = 2048, 5
TORCH_SEED = 20260928


def stats_inputs():
    """the fixed inputs of the three cases (shared with the tests through the fixtures themselves)"""
    s0 = np.repeat(PX.uniform_fill(4, (1, 10), 170, -1.0, 1.0), 4, axis=0)          # one state x 4 actions: Node.expand's shape (mcts.py:70-74)
    o = np.repeat(synth.make_frames(31, 1), 4, axis=0)
    start = PX.uniform_fill(4, (10,), 171, -1.0, 1.0)
    return s0, o, start


def load_reference_unpatched(weights):
    """the shimmed reference with torch's own noise (nothing patched; asserts that no earlier import patched it either)"""
    assert (F.dropout, torch.randn_like, torch.multinomial) == _ORIG, 'torch noise functions are patched in this process'
    sys.path.insert(0, REF)
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    from src.torchmodel import ActiveInferenceModel
    import src.torchmodel as TM
    model = ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, colour_channels=1, resolution=64)
    model.model_down.qs_net[9] = nn.Linear(576, 256)       # shim 1
    model.precision = torch.float32                        # shim 2
    for part, mod in (('top', model.model_top), ('mid', model.model_mid), ('down', model.model_down)):
        mod.load_state_dict({k[len(part) + 1:]: torch.from_numpy(v.copy()) for k, v in weights.items() if k.startswith(part + '.')})
    return model, TM


def main():
    torch.set_grad_enabled(False)
    weights = synth.make_weights(WSEED, GAIN)
    model, TM = load_reference_unpatched(weights)
    s0_np, o_np, start_np = stats_inputs()
    s0, o, start = torch.from_numpy(s0_np), torch.from_numpy(o_np), torch.from_numpy(start_np)
    pi = torch.eye(4)
    meta = dict(wseed=WSEED, gain=GAIN, torch_seed=TORCH_SEED)

    # recording wrapper: the reference's own entropy_bernoulli, its per-row sums remembered (no draw, no change of value)
    rec = []
    orig_eb = TM.entropy_bernoulli

    def entropy_bernoulli_rec(p, *a, **k):
        out = orig_eb(p, *a, **k)
        rec.append(torch.sum(out, dim=[1, 2, 3]).numpy().copy())
        return out
    TM.entropy_bernoulli = entropy_bernoulli_rec

    torch.manual_seed(TORCH_SEED)
    t = time.time()
    G = np.zeros((N_G, 4), np.float32); T = np.zeros((N_G, 5, 4), np.float32)       # term0, term1, term2, term2_1, term2_2
    for n in range(N_G):
        del rec[:]
        g, terms, _, _, _ = model.calculate_G(s0, pi, samples=S_G)
        assert len(rec) == 2 * S_G
        G[n] = g.numpy()
        for i in range(3):
            T[n, i] = terms[i].numpy()
        T[n, 3] = np.sum(np.stack(rec[0::2]), axis=0, dtype=np.float32) / np.float32(S_G)
        T[n, 4] = np.sum(np.stack(rec[1::2]), axis=0, dtype=np.float32) / np.float32(S_G)
    print(f'stats_calcG: {time.time() - t:.0f} s; G mean {G.mean(0)}, std {G.std(0)}; term2 std {T[:, 2].std(0)}', flush=True)
    np.savez_compressed(os.path.join(GOLD, 'stats_calcG.npz'), s0=s0_np, samples=S_G, G=G, t0=T[:, 0], t1=T[:, 1], t2=T[:, 2],
                        t2_1=T[:, 3], t2_2=T[:, 4], **meta)

    t = time.time()
    RG = np.zeros((N_R, 4), np.float32); RT = np.zeros((N_R, 3, 4), np.float32)
    for n in range(N_R):
        g, terms, _ = model.calculate_G_repeated(o, pi, steps=D_R, calc_mean=False, samples=S_R)
        RG[n] = g.numpy()
        for i in range(3):
            RT[n, i] = terms[i].numpy()
    print(f'stats_rollout: {time.time() - t:.0f} s; sum_G mean {RG.mean(0)}, std {RG.std(0)}', flush=True)
    np.savez_compressed(os.path.join(GOLD, 'stats_rollout.npz'), o=o_np, steps=D_R, samples=S_R, sum_G=RG, t0=RT[:, 0], t1=RT[:, 1], t2=RT[:, 2], **meta)
    TM.entropy_bernoulli = orig_eb

    t = time.time()
    SG = np.zeros(N_S, np.float32); SA = np.zeros((N_S, DEPTH_S), np.int8); Q = None
    for n in range(N_S):
        g, pi0, qpi = model.mcts_step_simulate(start, DEPTH_S, use_means=False)
        SG[n] = g
        SA[n] = pi0.numpy().argmax(1)
        assert np.array_equal(pi0.numpy().sum(1), np.ones(DEPTH_S))
        Q = qpi.numpy().copy() if Q is None else Q
        assert np.array_equal(Q, qpi.numpy())            # the habit net has no dropout: Qpi of step 0 is deterministic
    print(f'stats_simulate: {time.time() - t:.0f} s; G mean {SG.mean():.3f}, std {SG.std():.3f}; first action freq {np.bincount(SA[:, 0], minlength=4) / N_S}, Qpi {Q}', flush=True)
    np.savez_compressed(os.path.join(GOLD, 'stats_simulate.npz'), start=start_np, depth=DEPTH_S, G=SG, actions=SA, Qpi=Q, **meta)

    mpath = os.path.join(GOLD, 'MANIFEST.json')
    manifest = json.load(open(mpath))
    manifest['cases'].update({'stats_calcG': ['G', 's0', 'samples', 't0', 't1', 't2', 't2_1', 't2_2'],
                              'stats_rollout': ['o', 'steps', 'samples', 'sum_G', 't0', 't1', 't2'],
                              'stats_simulate': ['start', 'depth', 'G', 'actions', 'Qpi']})
    manifest['stats_cases'] = ('stats_*: oracle/make_golden_stats.py -- the shimmed reference under torch\'s OWN generator '
                               f'(torch.manual_seed({TORCH_SEED}), nothing patched); samples, compared as distributions')
    with open(mpath, 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
