"""ORACLE tooling -- runs ONLY in the build container (needs /root/reference).

The reference planner's EARLY STOP at the depth the benchmark runs (/root/reference/src/mcts.py:170-181: before every iteration,
`calc_threshold(normalization(root.N)) > threshold` ends the episode and `repeat` is returned), captured with the shim and the noise
injection of oracle/make_golden.py (imported, not repeated):

  mcts_deep_s10_thr     6 independent episodes x repeats = 50, Node.expand(samples=10), depth-5 simulations, threshold 0.5 = the
                        reference's default (mcts.py:140) and the threshold of bench.py's early-stop leg: the episodes stop at
                        iterations 41, 50, 50, 50, 29, 50 (`--probe`, below)
  mcts_deep_s10_thr04   the same episodes at threshold 0.4: stops at 28, 45, 19, 50, 19, 42
                        -- the lock-step planner's lagged host check, the compaction of stopped episodes and the device-side `active`
                        flags are compared with both (tests/test_gpu_parity.py).  Also stored: the stop statistic max(P) - mean(P)
                        of every check (`thr_stat`), so a test can see how close to the threshold an episode came.

Episode e draws its noise at global rows 4e+a (expansions), e (root encode, simulate steps) and e*depth+t (trajectory) -- what the
lock-step planner uses.  Fixtures hold tensors only.

Usage:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_thr            (about 2 minutes)
        PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_thr --probe    prints the stop statistic per iteration with the stop disabled
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
from oracle.make_golden import load_reference, GOLD
from oracle.make_golden_deep import capture
from oracle import synth

EPISODES, SAMPLES, REPEATS, DEPTH, STAGE0, FRAME_SEED = 6, 10, 50, 5, 800, 28
CASES = {'mcts_deep_s10_thr': 0.5, 'mcts_deep_s10_thr04': 0.4}            # chosen from the --probe table: see the assertion in capture_thr()


def capture_thr(name, thr, probe):
    torch.set_grad_enabled(False)
    WSEED, NSEED, gain = 1234, 7, 1.15
    model, inj, ref_mcts, state = load_reference(synth.make_weights(WSEED, gain), NSEED)
    meta = dict(wseed=WSEED, gain=gain, nseed=NSEED)
    stat = []                                        # the stop statistic of every check, in call order (a recording wrapper: returns the reference's value)
    orig_ct = ref_mcts.calc_threshold

    def calc_threshold_rec(P, axis):
        v = orig_ct(P, axis)
        stat.append(float(v))
        return v
    ref_mcts.calc_threshold = calc_threshold_rec
    try:
        keys, rep = capture(name, model, inj, ref_mcts, state, episodes=EPISODES, samples=SAMPLES, repeats=REPEATS, depth=DEPTH,
                            threshold=2.0 if probe else thr, stage0=STAGE0, frame_seed=FRAME_SEED, prior=False, use_habit=False, meta=meta)
    finally:
        ref_mcts.calc_threshold = orig_ct
    path = os.path.join(GOLD, name + '.npz')
    if probe:
        os.unlink(path)
        s = np.array(stat).reshape(EPISODES, REPEATS)
        for t in (0.3, 0.35, 0.4, 0.45, 0.5, 0.55):
            stops = [int(np.argmax(r > t)) if (r > t).any() else REPEATS for r in s]
            print('threshold', t, 'stops at', stops)
        return None, None
    g = dict(np.load(path))
    reps = [int(x) for x in g['repeats_done']]
    ts = np.full((EPISODES, REPEATS), np.nan, dtype=np.float32)          # one check per started iteration (+ the one that stops the episode)
    k = 0
    for e, r in enumerate(reps):
        n = r + 1 if r < REPEATS else REPEATS
        ts[e, :n] = stat[k:k + n]
        k += n
    assert k == len(stat)
    assert sum(10 <= r <= 45 for r in reps) >= 2 and any(r == REPEATS for r in reps) and len(set(reps)) >= 3, reps
    g['thr_stat'] = ts
    np.savez_compressed(path, **g)
    return sorted(keys + ['thr_stat']), rep


def main():
    if '--probe' in sys.argv:
        capture_thr('mcts_deep_s10_thr', 2.0, True)
        return
    mpath = os.path.join(GOLD, 'MANIFEST.json')
    manifest = json.load(open(mpath))
    for name, thr in CASES.items():
        keys, rep = capture_thr(name, thr, False)
        manifest['cases'][name] = keys
        print(name, json.dumps(rep))
    manifest['early_stop_case'] = 'mcts_deep_s10_thr, mcts_deep_s10_thr04: oracle/make_golden_thr.py (same shim and injection as make_golden.py)'
    with open(mpath, 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
