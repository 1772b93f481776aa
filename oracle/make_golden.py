"""ORACLE tooling -- runs ONLY in the build container (needs /root/reference).

Imports the reference (`/root/reference/src/torchmodel.py`, `src/mcts.py`) with the
three-line shim of SURVEY.md section 0 / appendix B:
    sys.modules['cv2'] = stub                       (mcts.py:5 imports cv2, unused)
    model.model_down.qs_net[9] = nn.Linear(576,256) (torchmodel.py:94 ships 256 -> raises)
    model.precision = torch.float32                 (torchmodel.py:355-359 undefined attr)
patches torch.nn.functional.dropout / torch.randn_like / torch.multinomial so the reference
consumes the addressable Philox stream of oracle/philox.py in its own draw order, loads the
synthetic weights of oracle/synth.py, runs the hot-path entry points and writes small
input/output fixtures to tests/golden/*.npz.  Nothing of the reference's source is copied:
fixtures hold tensors only.

Usage:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden
"""
import os
import sys
import types
import json
import numpy as np

sys.dont_write_bytecode = True
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import philox as PX
from oracle import synth
from oracle.efe_oracle import OracleModel, PhiloxNoise, categorical_from_uniform

REF = '/root/reference'
GOLD = os.path.join(ROOT, 'tests', 'golden')


# --------------------------------------------------------------------------
# plan-driven noise injection
# --------------------------------------------------------------------------
class Injector:
    """Queue of expected draws; the patched torch functions pop from it."""

    def __init__(self, seed):
        self.noise = PhiloxNoise(seed)
        self.q = []
        self.stage = 0

    # plan fragments -------------------------------------------------------
    def p_mask(self, tag0, nlayers, pas, sample, stage, ro, feats):
        for l in range(nlayers):
            self.q.append(('mask', tag0 + l, pas, sample, stage, ro, feats[l]))

    def p_trans(self, pas, sample, stage, ro, with_eps=True):
        self.p_mask(PX.TAG_MID, 3, pas, sample, stage, ro, [512] * 3)
        if with_eps:
            self.q.append(('eps', pas, sample, stage, ro))

    def p_dec(self, pas, sample, stage, ro):
        self.p_mask(PX.TAG_DEC, 4, pas, sample, stage, ro, [256, 256, 256, 16384])

    def p_enc(self, pas, sample, stage, ro, with_eps=True):
        self.p_mask(PX.TAG_ENC, 3, pas, sample, stage, ro, [256] * 3)
        if with_eps:
            self.q.append(('eps', pas, sample, stage, ro))

    def p_calculate_G(self, samples, stage, ro):
        """draw order of torchmodel.py:273-292"""
        for i in range(samples):
            self.p_trans(PX.PASS_T1, i, stage, ro)
            self.p_dec(PX.PASS_D1, i, stage, ro)
            self.p_enc(PX.PASS_E1, i, stage, ro)
        for j in range(samples):
            self.p_trans(PX.PASS_T2, j, stage, ro)
            self.p_dec(PX.PASS_D2A, j, stage, ro)
            self.q.append(('eps', PX.PASS_D2B, j, stage, ro))
            self.p_dec(PX.PASS_D2B, j, stage, ro)

    def p_trajectory(self, stage, ro):
        """torchmodel.py:332-347"""
        self.p_dec(PX.PASS_D1, 0, stage, ro)
        self.p_enc(PX.PASS_E1, 0, stage, ro)
        self.p_trans(PX.PASS_T2, 0, stage, ro)
        self.p_dec(PX.PASS_D2A, 0, stage, ro)
        self.q.append(('eps', PX.PASS_D2B, 0, stage, ro))
        self.p_dec(PX.PASS_D2B, 0, stage, ro)

    def p_simulate(self, depth, stage, episode):
        for t in range(depth):
            self.q.append(('uni', PX.PASS_HABIT, t, stage, episode))
            self.p_trans(PX.PASS_SIM, t, stage, episode)
        self.p_trajectory(stage, episode * depth)

    # patched torch entry points ------------------------------------------------
    def dropout(self, input, p=0.5, training=True, inplace=False):
        kind, tag, pas, sample, stage, ro, nf = self.q.pop(0)
        assert kind == 'mask' and input.shape[1] == nf and p == 0.5 and training, (kind, nf, tuple(input.shape))
        m = self.noise.mask(tag, input.shape[0], nf, pas, sample, stage, ro, fc4_perm=(nf == 16384))
        return input * m

    def randn_like(self, t, **kw):
        kind, pas, sample, stage, ro = self.q.pop(0)
        assert kind == 'eps', kind
        return self.noise.eps(t.shape[0], t.shape[1], pas, sample, stage, ro)

    def multinomial(self, probs, n, *a, **kw):
        kind, pas, sample, stage, ro = self.q.pop(0)
        assert kind == 'uni' and n == 1 and probs.dim() == 1
        u = self.noise.uniform(1, pas, sample, stage, ro)[0]
        a_, valid = categorical_from_uniform(probs.detach().numpy(), u)
        if not valid:
            raise RuntimeError('invalid multinomial probabilities')   # reference bare-except -> action 0
        return torch.tensor([a_])


def load_reference(weights, seed):
    sys.path.insert(0, REF)
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    import torch.nn.functional as F
    from src.torchmodel import ActiveInferenceModel
    import src.mcts as ref_mcts

    model = ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, colour_channels=1, resolution=64)
    model.model_down.qs_net[9] = nn.Linear(576, 256)       # shim 1
    model.precision = torch.float32                        # shim 2
    for part, mod in (('top', model.model_top), ('mid', model.model_mid), ('down', model.model_down)):
        sd = {k[len(part) + 1:]: torch.from_numpy(v.copy()) for k, v in weights.items() if k.startswith(part + '.')}
        mod.load_state_dict(sd)

    inj = Injector(seed)
    F.dropout = inj.dropout
    torch.randn_like = inj.randn_like
    torch.multinomial = inj.multinomial

    # wrap entry points so that each call pushes its own draw plan
    orig_G, orig_Gm, orig_sim = model.calculate_G, model.calculate_G_mean, model.mcts_step_simulate
    orig_rep, orig_rep4 = model.calculate_G_repeated, model.calculate_G_4_repeated
    state = {'ro': 0, 'episode': 0}

    def calculate_G(s0, pi0, samples=10):
        inj.p_calculate_G(samples, inj.stage, state['ro']); inj.stage += 1
        return orig_G(s0, pi0, samples=samples)

    def calculate_G_mean(s0, pi0):
        inj.p_calculate_G(1, inj.stage, state['ro']); inj.stage += 1
        return orig_Gm(s0, pi0)

    def calculate_G_repeated(o, pi, steps=1, calc_mean=False, samples=10):
        inj.p_enc(PX.PASS_ROOT, 0, inj.stage, state['ro'])
        return orig_rep(o, pi, steps=steps, calc_mean=calc_mean, samples=samples)

    def calculate_G_4_repeated(o, steps=1, calc_mean=False, samples=10):
        inj.p_enc(PX.PASS_ROOT, 0, inj.stage, state['ro'])
        return orig_rep4(o, steps=steps, calc_mean=calc_mean, samples=samples)

    def mcts_step_simulate(starting_s, depth, use_means=False):
        inj.p_simulate(depth, inj.stage, state['episode']); inj.stage += 1
        return orig_sim(starting_s, depth, use_means=use_means)

    model.calculate_G = calculate_G
    model.calculate_G_mean = calculate_G_mean
    model.calculate_G_repeated = calculate_G_repeated
    model.calculate_G_4_repeated = calculate_G_4_repeated
    model.mcts_step_simulate = mcts_step_simulate
    return model, inj, ref_mcts, state


def npy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().numpy().copy()
    return np.asarray(x)


def maxdiff(a, b):
    return float(np.max(np.abs(npy(a).astype(np.float64) - npy(b).astype(np.float64))))


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_grad_enabled(False)
    # (merged into an existing manifest: the other generators -- make_golden_deep / _invalid / _thr / _stats -- add their cases to the same
    # file, so the result does not depend on the order the generators are run in; oracle/regenerate_all.sh runs them all)
    mpath0 = os.path.join(GOLD, 'MANIFEST.json')
    manifest = json.load(open(mpath0)) if os.path.exists(mpath0) else {}
    manifest.update({'torch': torch.__version__, 'numpy': np.__version__,
                     'shim': ['cv2 stub', 'qs_net[9]=Linear(576,256)', 'precision=float32'],
                     'noise': 'Philox4x32-10 injected via F.dropout / torch.randn_like / torch.multinomial patches'})
    manifest.setdefault('cases', {})
    report = {}

    def save(name, **arrs):
        np.savez_compressed(os.path.join(GOLD, name + '.npz'), **{k: npy(v) for k, v in arrs.items()})
        manifest['cases'][name] = sorted(arrs.keys())

    WSEED = 1234
    for gain_tag, gain in (('g100', 1.0), ('g115', 1.15), ('g135', 1.35)):
        weights = synth.make_weights(WSEED, gain)
        NSEED = 7
        model, inj, ref_mcts, state = load_reference(weights, NSEED)
        orc = OracleModel(weights, PhiloxNoise(NSEED))
        meta = dict(wseed=WSEED, gain=gain, nseed=NSEED)

        # ---------------- networks ----------------------------
        M = 5
        frames = torch.from_numpy(synth.make_frames(11, M))
        s = torch.from_numpy(PX.uniform_fill(3, (M, 10), 50, -1.5, 1.5))
        pi = torch.eye(4)[torch.tensor([0, 1, 2, 3, 1])]
        stage = 40
        inj.p_trans(PX.PASS_T1, 0, stage, 0)
        t_ps1, t_mean, t_lv = model.model_mid.transition_with_sample(pi, s)
        inj.p_dec(PX.PASS_D1, 0, stage, 0)
        d_po = model.model_down.decoder(s)
        inj.p_enc(PX.PASS_E1, 0, stage, 0)
        e_s, e_mean, e_lv = model.model_down.encoder_with_sample(frames)
        h_logits, h_q, h_logq = model.model_top.encode_s(s)
        assert not inj.q
        o_ps1, o_mean, o_lv = orc.transition_with_sample(pi, s, PX.PASS_T1, 0, stage)
        o_po = orc.decoder(s, PX.PASS_D1, 0, stage)
        o_es, o_emean, o_elv = orc.encoder_with_sample(frames, PX.PASS_E1, 0, stage)
        o_hl, o_hq, o_hlq = orc.encode_s(s)
        report[f'nets_{gain_tag}'] = dict(trans=maxdiff(t_ps1, o_ps1), dec=maxdiff(d_po, o_po),
                                          enc=max(maxdiff(e_mean, o_emean), maxdiff(e_lv, o_elv), maxdiff(e_s, o_es)),
                                          habit=maxdiff(h_q, o_hq))
        if gain_tag == 'g115':
            from src import torchutils as TU
            p = torch.from_numpy(PX.uniform_fill(5, (3, 1, 64, 64), 60, 0.0, 1.0))
            p[0, 0, :4, :4] = 0.0
            p[0, 0, 4:8, :4] = 1.0
            save('helpers', p=p, lv=t_lv, ent_normal=TU.entropy_normal_from_logvar(t_lv),
                 ent_bern=TU.entropy_bernoulli(p), reward=model.check_reward(p),
                 ent_bern_sum=torch.sum(TU.entropy_bernoulli(p), dim=[1, 2, 3]))
        save(f'nets_{gain_tag}', frames=frames, s=s, pi=pi, stage=stage,
             t_ps1=t_ps1, t_mean=t_mean, t_lv=t_lv, d_po=d_po, e_s=e_s, e_mean=e_mean, e_lv=e_lv,
             h_logits=h_logits, h_q=h_q, h_logq=h_logq, **meta)

        # ---------------- calculate_G (8a-8), calculate_G_mean (8a-9) -------------
        for tag, (M, S) in (('m4s1', (4, 1)), ('m6s3', (6, 3))):
            s0 = torch.from_numpy(PX.uniform_fill(4, (M, 10), 51 + M, -1.0, 1.0))
            pi0 = torch.eye(4)[torch.arange(M) % 4]
            inj.stage = 10
            G, terms, ps1, ps1_mean, po1 = model.calculate_G(s0, pi0, samples=S)
            assert not inj.q
            oG, oterms, ops1, ops1m, opo1 = orc.calculate_G(s0, pi0, S, 10)
            report[f'G_{tag}_{gain_tag}'] = dict(G=maxdiff(G, oG), ps1=maxdiff(ps1, ops1), po1=maxdiff(po1, opo1),
                                                 t2_1=float(orc.last_term2_parts[0].abs().max()))
            save(f'calcG_{tag}_{gain_tag}', s0=s0, pi0=pi0, samples=S, stage=10, G=G, t0=terms[0], t1=terms[1], t2=terms[2],
                 t2_1=orc.last_term2_parts[0], t2_2=orc.last_term2_parts[1],
                 ps1=ps1, ps1_mean=ps1_mean, po1=po1, **meta)
        s0 = torch.from_numpy(PX.uniform_fill(4, (4, 10), 70, -1.0, 1.0)[:1].repeat(4, 0))
        inj.stage = 20
        G, terms, ps1_mean, po1 = model.calculate_G_mean(s0, model.pi_one_hot)
        oG, oterms, ops1m, opo1 = orc.calculate_G_mean(s0, orc.pi_one_hot, 20)
        report[f'Gmean_{gain_tag}'] = dict(G=maxdiff(G, oG), ps1m=maxdiff(ps1_mean, ops1m))
        save(f'calcGmean_{gain_tag}', s0=s0, stage=20, G=G, t0=terms[0], t1=terms[1], t2=terms[2],
             t2_1=orc.last_term2_parts[0], ps1_mean=ps1_mean, po1=po1, **meta)

        if gain_tag != 'g115':
            continue

        # ---------------- calculate_G_repeated / _4_repeated (8a-10) ---------------
        for tag, (M, D, S, cm) in (('cfg1', (4, 1, 1, False)), ('m8d2s2', (8, 2, 2, False)), ('m8d2s2mean', (8, 2, 2, True))):
            fr = synth.make_frames(21, (M + 3) // 4)
            o = torch.from_numpy(np.repeat(fr, 4, axis=0)[:M])          # util.py:56-57 intent: row 4i+a
            pi = torch.eye(4).repeat((M + 3) // 4, 1)[:M]               # util.py:59-60
            inj.stage = 30
            sum_G, sum_terms, po1 = model.calculate_G_repeated(o, pi, steps=D, calc_mean=cm, samples=S)
            assert not inj.q
            oG, oT, opo1 = orc.calculate_G_repeated(o, pi, D, cm, S, 30)
            report[f'rep_{tag}'] = dict(G=maxdiff(sum_G, oG), po1=maxdiff(po1, opo1))
            from src.util import softmax_multi_with_log
            Ppi, logPpi = softmax_multi_with_log(-sum_G.numpy(), 4)
            save(f'rollout_{tag}', o=o, pi=pi, steps=D, samples=S, calc_mean=cm, stage=30, sum_G=sum_G,
                 t0=sum_terms[0], t1=sum_terms[1], t2=sum_terms[2], po1=po1, Ppi=Ppi, logPpi=logPpi, **meta)
        for tag, cm in (('s2', False), ('mean', True)):
            o = torch.from_numpy(np.repeat(synth.make_frames(22, 1), 4, axis=0))
            inj.stage = 35
            sum_G, sum_terms, po1 = model.calculate_G_4_repeated(o, steps=2, calc_mean=cm, samples=2)
            assert not inj.q
            oG, oT, opo1 = orc.calculate_G_4_repeated(o, 2, cm, 2, 35)
            report[f'rep4_{tag}'] = dict(G=maxdiff(sum_G, oG))
            save(f'rollout4_{tag}', o=o, steps=2, samples=2, calc_mean=cm, stage=35, sum_G=sum_G,
                 t0=sum_terms[0], t1=sum_terms[1], t2=sum_terms[2], po1=po1, **meta)

        # ---------------- mcts_step_simulate / given_trajectory (8a-11) ------------
        for tag, um in (('sample', False), ('means', True)):
            start = torch.from_numpy(PX.uniform_fill(4, (10,), 80, -1.0, 1.0))
            inj.stage = 50
            state['episode'] = 3
            Gs, pi0, Qpi = model.mcts_step_simulate(start, 3, use_means=um)
            assert not inj.q
            oGs, opi0, oQpi = orc.mcts_step_simulate(start, 3, um, 50, episode=3)
            report[f'sim_{tag}'] = dict(G=abs(Gs - oGs), pi=maxdiff(pi0, opi0), q=maxdiff(Qpi, oQpi))
            save(f'simulate_{tag}', start=start, depth=3, use_means=um, stage=50, episode=3, G=Gs, pi0=pi0, Qpi=Qpi, **meta)
        state['episode'] = 0

        # ---------------- active_inference_mcts (8a-12) ---------------------------
        for tag, um in (('means', True), ('samples', False)):
            params = ref_mcts.MCTS_Params()
            params.repeats = 6
            params.simulation_depth = 3
            params.use_means = um
            params.threshold = 0.8
            frame = torch.from_numpy(synth.make_frames(23, 1)[0, 0][:, :, None].copy())   # HWC [64,64,1]
            inj.stage = 100
            inj.p_enc(PX.PASS_ROOT, 0, inj.stage, 0, with_eps=False); inj.stage += 1     # root encode mcts.py:158
            ref_mcts.NODE_ID = 0
            path, reps, explored, all_paths, all_G = ref_mcts.active_inference_mcts(model, frame, params, o_shape=(1, 64, 64))
            assert not inj.q
            ap = np.full((len(all_paths), 8), -1, dtype=np.int64)
            for i, p_ in enumerate(all_paths):
                ap[i, :len(p_)] = [int(x) for x in p_]
            save(f'mcts_{tag}', frame=frame, repeats=6, simulation_depth=3, use_means=um, threshold=0.8, stage=100,
                 final_path=np.asarray(path, dtype=np.int64), repeats_done=reps, states_explored=explored,
                 all_paths=ap, all_paths_G=np.asarray(all_G, dtype=np.float64), **meta)
            report[f'mcts_{tag}'] = dict(path=[int(x) for x in path], reps=int(reps), n_paths=len(all_paths))

        # ---------------- tree policy with the habit prior in the exploration bonus (mcts.py:44-45) ---------------
        params = ref_mcts.MCTS_Params()
        params.repeats, params.simulation_depth, params.use_means, params.threshold = 7, 3, True, 0.9
        params.using_prior_for_exploration = True
        frame = torch.from_numpy(synth.make_frames(24, 1)[0, 0][:, :, None].copy())
        inj.stage = 200
        inj.p_enc(PX.PASS_ROOT, 0, inj.stage, 0, with_eps=False); inj.stage += 1
        path, reps, explored, all_paths, all_G = ref_mcts.active_inference_mcts(model, frame, params, o_shape=(1, 64, 64))
        assert not inj.q
        ap = np.full((len(all_paths), 10), -1, dtype=np.int64)
        for i, p_ in enumerate(all_paths):
            ap[i, :len(p_)] = [int(x) for x in p_]
        save('mcts_prior', frame=frame, repeats=7, simulation_depth=3, use_means=True, threshold=0.9, stage=200,
             final_path=np.asarray(path, dtype=np.int64), repeats_done=reps, states_explored=explored,
             all_paths=ap, all_paths_G=np.asarray(all_G, dtype=np.float64), **meta)
        report['mcts_prior'] = dict(path=[int(x) for x in path], reps=int(reps), n_paths=len(all_paths))

        # ---------------- BASELINE configs[2] shape: Node.expand(samples=10), 8 independent episodes ---------------
        # The reference planner hard-wires expand(samples=1) (mcts.py:172,184) although Node.expand takes `samples` (mcts.py:64):
        # the default is patched to 10 here, nothing else.  Episode e draws its noise at global rows 4e+a (expansions), e
        # (root encode, simulate steps) and e*depth+t (trajectory), exactly what the lock-step planner uses.
        E_B, S_B, REP_B, DEP_B, THR_B = 8, 10, 8, 5, 0.3
        orig_expand = ref_mcts.Node.expand
        orig_init = ref_mcts.Node.__init__
        created = []

        def expand10(self, use_means=False, samples=S_B):
            return orig_expand(self, use_means=use_means, samples=samples)

        def init_capture(self, *a, **k):
            orig_init(self, *a, **k)
            created.append(self)
        ref_mcts.Node.expand = expand10
        ref_mcts.Node.__init__ = init_capture
        frames_b = synth.make_frames(25, E_B)
        fp = np.full((E_B, REP_B + 2), -1, dtype=np.int64); apb = np.full((E_B, REP_B, REP_B + 2), -1, dtype=np.int64)
        agb = np.zeros((E_B, REP_B), dtype=np.float64); npaths = np.zeros(E_B, dtype=np.int64)
        repd = np.zeros(E_B, dtype=np.int64); expl = np.zeros(E_B, dtype=np.int64); rootN = np.zeros((E_B, 4), dtype=np.float32)
        for e in range(E_B):
            params = ref_mcts.MCTS_Params()
            params.repeats, params.simulation_depth, params.use_means, params.threshold = REP_B, DEP_B, False, THR_B
            state['ro'], state['episode'] = 4 * e, e
            inj.stage = 300
            inj.p_enc(PX.PASS_ROOT, 0, inj.stage, e, with_eps=False); inj.stage += 1
            del created[:]
            frame = torch.from_numpy(frames_b[e, 0][:, :, None].copy())
            path, reps, explored, all_paths, all_G = ref_mcts.active_inference_mcts(model, frame, params, o_shape=(1, 64, 64))
            assert not inj.q
            fp[e, :len(path)] = [int(x) for x in path]
            for i, p_ in enumerate(all_paths):
                apb[e, i, :len(p_)] = [int(x) for x in p_]
            agb[e, :len(all_G)] = all_G
            npaths[e], repd[e], expl[e] = len(all_paths), reps, explored
            rootN[e] = created[0].N.numpy()
        ref_mcts.Node.expand = orig_expand
        ref_mcts.Node.__init__ = orig_init
        state['ro'], state['episode'] = 0, 0
        save('mcts_batch_s10', frames=frames_b, episodes=E_B, samples=S_B, repeats=REP_B, simulation_depth=DEP_B, threshold=THR_B,
             stage=300, final_path=fp, all_paths=apb, all_paths_G=agb, n_paths=npaths, repeats_done=repd, states_explored=expl,
             root_N=rootN, **meta)
        report['mcts_batch_s10'] = dict(reps=[int(x) for x in repd], n_paths=[int(x) for x in npaths])

    # ---------------- the reference's resolution-32 variant (pi 3, 3 x 32 x 32, last_strides = 1) -------------------
    # Its NETWORKS are constructible and runnable (with the same kind of shim as the dSprites model: the first encoder Linear
    # ships 256 inputs, torchmodel.py:94, where the trunk emits 64); its calculate_G is not (calc_reward_animalai is undefined,
    # torchmodel.py:213-214).  So: network-level fixtures only; the EFE terms of this geometry stay build-defined.
    if True:
        sys.path.insert(0, REF)
        sys.modules.setdefault('cv2', types.ModuleType('cv2'))
        from src.torchmodel import ActiveInferenceModel
        A32, C32, R32, NSEED = 3, 3, 32, 7
        weights = synth.make_weights(WSEED, 1.15, A32, C32, R32)
        ref = ActiveInferenceModel(10, A32, 0.0, 1.0, 1.0, colour_channels=C32, resolution=R32)
        ref.model_down.qs_net[9] = nn.Linear(64, 256)                      # shim (as 576 for resolution 64)
        for part, mod in (('top', ref.model_top), ('mid', ref.model_mid), ('down', ref.model_down)):
            mod.load_state_dict({k[len(part) + 1:]: torch.from_numpy(v.copy()) for k, v in weights.items() if k.startswith(part + '.')})
        inj = Injector(NSEED)
        import torch.nn.functional as F
        F.dropout = inj.dropout
        torch.randn_like = inj.randn_like
        M, stage = 5, 60
        s = torch.from_numpy(PX.uniform_fill(3, (M, 10), 90, -1.5, 1.5))
        pi = torch.eye(A32)[torch.tensor([0, 1, 2, 1, 0])]
        frames = torch.from_numpy(synth.make_frames_rgb(15, M, C32, R32))
        inj.p_trans(PX.PASS_T1, 0, stage, 0)
        t_ps1, t_mean, t_lv = ref.model_mid.transition_with_sample(pi, s)
        inj.q.extend(('mask', PX.TAG_DEC + l, PX.PASS_D1, 0, stage, 0, f) for l, f in enumerate([256, 256, 256, 16384]))
        d_po = ref.model_down.decoder(s)
        inj.p_enc(PX.PASS_E1, 0, stage, 0)
        e_s, e_mean, e_lv = ref.model_down.encoder_with_sample(frames)
        h_logits, h_q, h_logq = ref.model_top.encode_s(s)
        assert not inj.q
        orc = OracleModel(weights, PhiloxNoise(NSEED), pi_dim=A32, channels=C32, resolution=R32)
        o_po = orc.decoder(s, PX.PASS_D1, 0, stage)
        o_es, o_em, o_elv = orc.encoder_with_sample(frames, PX.PASS_E1, 0, stage)
        report['nets32_g115'] = dict(dec=maxdiff(d_po, o_po), enc=max(maxdiff(e_mean, o_em), maxdiff(e_s, o_es)),
                                     trans=maxdiff(t_ps1, orc.transition_with_sample(pi, s, PX.PASS_T1, 0, stage)[0]))
        save('nets32_g115', frames=frames, s=s, pi=pi, stage=stage, t_ps1=t_ps1, t_mean=t_mean, t_lv=t_lv, d_po=d_po, e_s=e_s, e_mean=e_mean,
             e_lv=e_lv, h_logits=h_logits, h_q=h_q, h_logq=h_logq, wseed=WSEED, gain=1.15, nseed=NSEED, pi_dim=A32, channels=C32, resolution=R32)

    with open(os.path.join(GOLD, 'MANIFEST.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(json.dumps(report, indent=1))


if __name__ == '__main__':
    main()
