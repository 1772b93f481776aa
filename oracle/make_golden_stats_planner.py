"""ORACLE tooling -- runs ONLY in the build container (needs /root/reference).

The statistical pin of oracle/make_golden_stats.py one level up: whole DECISIONS of the reference planner under the reference's own random
number generator.  N_D calls of `active_inference_mcts(model, frame, params)` (/root/reference/src/mcts.py:150-195) on one fixed frame with
the shimmed reference, NOTHING of torch's noise patched, `torch.manual_seed`: the reference's default parameters (use_means = True, C = 1)
at repeats = 12, simulation_depth = 3 and threshold 0.3 (at the default 0.5 only 2 of 512 such decisions stop early; at 0.3 a third do).  Stored per decision: the final path (after the opposite-pair trimming), repeats_done
(the early stop), the root's visit counts and -- for the first action -- the untrimmed argmax of the root visits.

  tests/golden/stats_planner.npz  <-  compared with the lock-step planner in device-noise mode (2 048 episodes on the same frame: every
  episode draws at its own global rows) as categorical distributions (tests/test_noise_statistics.py).  Fixtures hold tensors only.

Usage:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_stats_planner        (about 5 minutes on 8 cores)
"""
import json
import os
import sys
import time
import types

import numpy as np

sys.dont_write_bytecode = True
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import synth
from oracle.make_golden_stats import load_reference_unpatched, GOLD, WSEED, GAIN

N_D, REPEATS, DEPTH, THRESHOLD, FRAME_SEED, TORCH_SEED = 512, 12, 3, 0.3, 33, 20260929


def main():
    torch.set_grad_enabled(False)
    model, _ = load_reference_unpatched(synth.make_weights(WSEED, GAIN))
    import src.mcts as ref_mcts
    created = []
    orig_init = ref_mcts.Node.__init__

    def init_capture(self, *a, **k):
        orig_init(self, *a, **k)
        created.append(self)
    ref_mcts.Node.__init__ = init_capture
    params = ref_mcts.MCTS_Params()                      # the reference's defaults: use_means True, C 1.0, simulation_repeats 1
    params.repeats, params.simulation_depth, params.threshold = REPEATS, DEPTH, THRESHOLD
    frame_np = synth.make_frames(FRAME_SEED, 1)
    frame = torch.from_numpy(frame_np[0, 0][:, :, None].copy())          # HWC [64, 64, 1], as the environment emits it
    paths = np.full((N_D, REPEATS + 2), -1, np.int8); plen = np.zeros(N_D, np.int8)
    reps = np.zeros(N_D, np.int8); rootN = np.zeros((N_D, 4), np.float32); explored = np.zeros(N_D, np.int16)
    torch.manual_seed(TORCH_SEED)
    t = time.time()
    try:
        for n in range(N_D):
            del created[:]
            path, r, ex, _, _ = ref_mcts.active_inference_mcts(model, frame, params, o_shape=(1, 64, 64))
            paths[n, :len(path)] = [int(x) for x in path]; plen[n] = len(path)
            reps[n], explored[n] = r, ex
            rootN[n] = created[0].N.numpy()
    finally:
        ref_mcts.Node.__init__ = orig_init
    print(f'stats_planner: {time.time() - t:.0f} s; repeats_done histogram {np.bincount(reps, minlength=REPEATS + 1)}; '
          f'argmax root visits {np.bincount(rootN.argmax(1), minlength=4) / N_D}; path length histogram {np.bincount(plen)}', flush=True)
    np.savez_compressed(os.path.join(GOLD, 'stats_planner.npz'), frame=frame_np, repeats=REPEATS, simulation_depth=DEPTH, threshold=params.threshold,
                        use_means=int(params.use_means), C=params.C, paths=paths, path_len=plen, repeats_done=reps, states_explored=explored, root_N=rootN,
                        wseed=WSEED, gain=GAIN, torch_seed=TORCH_SEED)
    mpath = os.path.join(GOLD, 'MANIFEST.json')
    manifest = json.load(open(mpath))
    manifest['cases']['stats_planner'] = ['frame', 'repeats', 'simulation_depth', 'threshold', 'use_means', 'C', 'paths', 'path_len', 'repeats_done', 'states_explored', 'root_N']
    manifest['stats_planner_case'] = f'stats_planner: oracle/make_golden_stats_planner.py -- {N_D} decisions of the reference planner under torch\'s own generator (torch.manual_seed({TORCH_SEED}))'
    with open(mpath, 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
