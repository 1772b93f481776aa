"""ORACLE tooling -- runs ONLY in the build container (needs /root/reference).

Planner fixtures at the depth the benchmark runs (BASELINE configs[2]: 50 expansions, 10 MC samples per expansion, simulation
depth 5), captured from the reference planner `/root/reference/src/mcts.py:150-195` with the shim and the noise injection of
oracle/make_golden.py (imported, not repeated):

  mcts_deep_s10   3 independent episodes x repeats = 50 (early stop disabled: every episode runs all 50 iterations, trees three or
                  more levels deep, long paths through the trimming of mcts.py:110-126), Node.expand(samples=10), depth-5 simulations
  mcts_prior_s10  using_prior_for_exploration (mcts.py:44-45) TOGETHER with Node.expand(samples=10) and use_habit (mcts.py:166-169;
                  the shortcut is evaluated, not taken), 2 episodes x 12 iterations

Episode e draws its noise at global rows 4e+a (expansions), e (root encode, simulate steps) and e*depth+t (trajectory) -- what the
lock-step planner uses -- so both `active_inference_mcts` (episode alone) and `active_inference_mcts_batch` are compared with it.
Fixtures hold tensors only.     Usage:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_deep
"""
import json
import os
import sys
import time

import numpy as np

sys.dont_write_bytecode = True
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import philox as PX
from oracle import synth
from oracle.make_golden import load_reference, GOLD


def capture(name, model, inj, ref_mcts, state, *, episodes, samples, repeats, depth, threshold, stage0, frame_seed, prior, use_habit, meta):
    orig_expand, orig_init = ref_mcts.Node.expand, ref_mcts.Node.__init__
    created = []

    def expand_s(self, use_means=False, samples=samples):        # the reference planner hard-wires expand(samples=1) (mcts.py:172,184)
        return orig_expand(self, use_means=use_means, samples=samples)

    def init_capture(self, *a, **k):
        orig_init(self, *a, **k)
        created.append(self)
    ref_mcts.Node.expand, ref_mcts.Node.__init__ = expand_s, init_capture
    E, R = episodes, repeats
    frames = synth.make_frames(frame_seed, E)
    fp = np.full((E, R + 2), -1, dtype=np.int64); ap = np.full((E, R, R + 2), -1, dtype=np.int64)
    ag = np.zeros((E, R), dtype=np.float64); npaths = np.zeros(E, dtype=np.int64)
    repd = np.zeros(E, dtype=np.int64); expl = np.zeros(E, dtype=np.int64); rootN = np.zeros((E, 4), dtype=np.float32)
    nodes = np.zeros(E, dtype=np.int64); maxlen = np.zeros(E, dtype=np.int64)
    try:
        for e in range(E):
            params = ref_mcts.MCTS_Params()
            params.repeats, params.simulation_depth, params.use_means, params.threshold = R, depth, False, threshold
            params.using_prior_for_exploration, params.use_habit = prior, use_habit
            state['ro'], state['episode'] = 4 * e, e
            inj.stage = stage0
            inj.p_enc(PX.PASS_ROOT, 0, inj.stage, e, with_eps=False); inj.stage += 1
            del created[:]
            frame = torch.from_numpy(frames[e, 0][:, :, None].copy())
            t = time.time()
            path, reps, explored, all_paths, all_G = ref_mcts.active_inference_mcts(model, frame, params, o_shape=(1, 64, 64))
            assert not inj.q
            fp[e, :len(path)] = [int(x) for x in path]
            for i, p_ in enumerate(all_paths):
                ap[e, i, :len(p_)] = [int(x) for x in p_]
            ag[e, :len(all_G)] = all_G
            npaths[e], repd[e], expl[e] = len(all_paths), reps, explored
            rootN[e] = created[0].N.numpy()
            nodes[e] = len(created); maxlen[e] = max(len(p_) for p_ in all_paths)
            print(f'{name} episode {e}: {time.time() - t:.1f} s, reps {reps}, nodes {len(created)}, longest path {maxlen[e]}', flush=True)
    finally:
        ref_mcts.Node.expand, ref_mcts.Node.__init__ = orig_expand, orig_init
        state['ro'], state['episode'] = 0, 0
    arrs = dict(frames=frames, episodes=E, samples=samples, repeats=R, simulation_depth=depth, threshold=threshold, stage=stage0,
                using_prior_for_exploration=int(prior), use_habit=int(use_habit), final_path=fp, all_paths=ap, all_paths_G=ag,
                n_paths=npaths, repeats_done=repd, states_explored=expl, root_N=rootN, n_nodes=nodes, **meta)
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()})
    return sorted(arrs.keys()), dict(reps=[int(x) for x in repd], nodes=[int(x) for x in nodes], longest_path=[int(x) for x in maxlen])


def main():
    torch.set_grad_enabled(False)
    WSEED, NSEED, gain = 1234, 7, 1.15
    weights = synth.make_weights(WSEED, gain)
    model, inj, ref_mcts, state = load_reference(weights, NSEED)
    meta = dict(wseed=WSEED, gain=gain, nseed=NSEED)
    report, cases = {}, {}
    cases['mcts_deep_s10'], report['mcts_deep_s10'] = capture('mcts_deep_s10', model, inj, ref_mcts, state, episodes=3, samples=10, repeats=50,
                                                              depth=5, threshold=2.0, stage0=400, frame_seed=26, prior=False, use_habit=False, meta=meta)
    cases['mcts_prior_s10'], report['mcts_prior_s10'] = capture('mcts_prior_s10', model, inj, ref_mcts, state, episodes=2, samples=10, repeats=12,
                                                                depth=5, threshold=0.45, stage0=600, frame_seed=27, prior=True, use_habit=True, meta=meta)
    # ---- the upstream-intent reward: the reference's own calc_reward / check_reward formula applied to the NHWC view
    # of an image batch -- what the upstream TensorFlow code computes -- as the pin of oracle.efe_oracle.check_reward_upstream_intent
    from src import torchutils as TU
    p_img = torch.from_numpy(PX.uniform_fill(5, (3, 1, 64, 64), 60, 0.0, 1.0))
    p_img[0, 0, :4, :4] = 0.0
    p_img[0, 0, 4:8, :4] = 1.0
    r_intent = torch.mean(TU.calc_reward(p_img.permute(0, 2, 3, 1)), dim=[1, 2, 3]) * 10.0
    np.savez_compressed(os.path.join(GOLD, 'helpers_intent.npz'), p=p_img.numpy(), reward_upstream_intent=r_intent.numpy())
    cases['helpers_intent'] = ['p', 'reward_upstream_intent']
    mpath = os.path.join(GOLD, 'MANIFEST.json')
    manifest = json.load(open(mpath))
    manifest['cases'].update(cases)
    manifest['deep_planner_cases'] = 'mcts_deep_s10, mcts_prior_s10: oracle/make_golden_deep.py (same shim and injection as make_golden.py)'
    with open(mpath, 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(json.dumps(report, indent=1))


if __name__ == '__main__':
    main()
