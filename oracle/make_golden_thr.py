"""ORACLE tooling -- runs ONLY in the build container (needs /root/reference).

The reference planner's EARLY STOP at the depth the benchmark runs (/root/reference/src/mcts.py:170-181: before every iteration,
`calc_threshold(normalization(root.N)) > threshold` ends the episode and `repeat` is returned), captured with the shim and the noise
injection of oracle/make_golden.py (imported, not repeated):

  mcts_deep_s10_thr   6 independent episodes x repeats = 50, Node.expand(samples=10), depth-5 simulations, threshold THR chosen (with
                      `--probe`, below) so that the episodes stop at different iterations: at least two between 10 and 45 and at
                      least one runs all 50 -- the lock-step planner's lagged host check, compaction of stopped episodes and the
                      device-side `active` flags are compared with it (tests/test_gpu_parity.py), and bench.py's threshold-0.5 leg
                      names it.  Also stored: the stop statistic max(P) - mean(P) of every iteration (`thr_stat`), so a test can
                      see how close to the threshold an episode came.

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
THR = 0.4            # chosen from the --probe table: see the assertion in main()


def main():
    torch.set_grad_enabled(False)
    WSEED, NSEED, gain = 1234, 7, 1.15
    probe = '--probe' in sys.argv
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
        keys, rep = capture('mcts_deep_s10_thr', model, inj, ref_mcts, state, episodes=EPISODES, samples=SAMPLES, repeats=REPEATS, depth=DEPTH,
                            threshold=2.0 if probe else THR, stage0=STAGE0, frame_seed=FRAME_SEED, prior=False, use_habit=False, meta=meta)
    finally:
        ref_mcts.calc_threshold = orig_ct
    path = os.path.join(GOLD, 'mcts_deep_s10_thr.npz')
    if probe:
        os.unlink(path)
        s = np.array(stat).reshape(EPISODES, REPEATS)
        for thr in (0.3, 0.35, 0.4, 0.45, 0.5, 0.55):
            stops = [int(np.argmax(r > thr)) if (r > thr).any() else REPEATS for r in s]
            print('threshold', thr, 'stops at', stops)
        return
    g = dict(np.load(path))
    reps = [int(x) for x in g['repeats_done']]
    # one check per started iteration (+ the one that stops the episode)
    ts = np.full((EPISODES, REPEATS), np.nan, dtype=np.float32)
    k = 0
    for e, r in enumerate(reps):
        n = r + 1 if r < REPEATS else REPEATS
        ts[e, :n] = stat[k:k + n]
        k += n
    assert k == len(stat)
    assert sum(10 <= r <= 45 for r in reps) >= 2 and any(r == REPEATS for r in reps) and len(set(reps)) >= 4, reps
    g['thr_stat'] = ts
    np.savez_compressed(path, **g)
    mpath = os.path.join(GOLD, 'MANIFEST.json')
    manifest = json.load(open(mpath))
    manifest['cases']['mcts_deep_s10_thr'] = sorted(keys + ['thr_stat'])
    manifest['early_stop_case'] = 'mcts_deep_s10_thr: oracle/make_golden_thr.py (same shim and injection as make_golden.py)'
    with open(mpath, 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(json.dumps(rep, indent=1))


if __name__ == '__main__':
    main()
