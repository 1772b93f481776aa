"""dev tool: observed |engine - fixture| per quantity over the committed goldens (basis of the tolerances in tests/test_gpu_parity.py)

  python tools/measure_errors.py [bf16x3 | f16x2]      with the opt-in split-operand experiment mfma_bf16x3 / mfma_f16x2 on (csrc/bf16x3.hip);
                                                       the small fixtures reach k_fc4_b3 only -- the last block measures the large-launch kernels too"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import daimc_amd
from conftest import load_golden, eps_calcG, eps_rollout
from oracle import synth, philox as PX
worst = {}
OPT = {'bf16x3': 'mfma_bf16x3', 'f16x2': 'mfma_f16x2'}.get(sys.argv[1] if len(sys.argv) > 1 else '', None)
B3 = OPT is not None
_orig_load = daimc_amd.ActiveInferenceModel.load_flat_weights
def _load(self, w):
    _orig_load(self, w)
    if B3:
        self.set_option(OPT, 1)
daimc_amd.ActiveInferenceModel.load_flat_weights = _load
def upd(k, a, b):
    d = float(np.max(np.abs(a.detach().cpu().numpy() - b)))
    worst[k] = max(worst.get(k, 0.0), d)
for gain in ('g100', 'g115', 'g135'):
    for case in ('m4s1', 'm6s3'):
        g = load_golden(f'calcG_{case}_{gain}')
        m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=int(g['nseed']), init_weights=False)
        m.load_flat_weights(synth.make_weights(int(g['wseed']), float(g['gain'])))
        S, st, M = int(g['samples']), int(g['stage']), len(g['s0'])
        parts = []
        G, T, ps1, ps1m, po1 = m.calculate_G(g['s0'], g['pi0'], samples=S, stage=st, eps=eps_calcG(int(g['nseed']), M, S, st), _parts=parts)
        upd('term0', T[0], g['t0']); upd('term1', T[1], g['t1']); upd('term2', T[2], g['t2']); upd('G', G, g['G'])
        upd('t2_1', parts[0][0], g['t2_1']); upd('po1', po1, g['po1']); upd('ps1', ps1, g['ps1'])
    g = load_golden(f'nets_{gain}')
    m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=int(g['nseed']), init_weights=False)
    m.load_flat_weights(synth.make_weights(int(g['wseed']), float(g['gain'])))
    st, seed, M = int(g['stage']), int(g['nseed']), len(g['s'])
    ps1, mean, lv = m.model_mid.transition_with_sample(g['pi'], g['s'], stage=st, pass_=PX.PASS_T1, eps=PX.normals(seed, M, 10, PX.PASS_T1, 0, st))
    upd('trans mean', mean, g['t_mean']); upd('trans logvar', lv, g['t_lv'])
    upd('decoder po', m.model_down.decoder(g['s'], stage=st, pass_=PX.PASS_D1), g['d_po'])
    s, em, elv = m.model_down.encoder_with_sample(g['frames'], stage=st, pass_=PX.PASS_E1, eps=PX.normals(seed, M, 10, PX.PASS_E1, 0, st))
    upd('enc mean', em, g['e_mean']); upd('enc logvar', elv, g['e_lv'])
for name in ('rollout_cfg1', 'rollout_m8d2s2', 'rollout_m8d2s2mean'):
    g = load_golden(name)
    m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=int(g['nseed']), init_weights=False)
    m.load_flat_weights(synth.make_weights(int(g['wseed']), float(g['gain'])))
    D, S, st, M = int(g['steps']), int(g['samples']), int(g['stage']), len(g['o'])
    sG, T, po1 = m.calculate_G_repeated(g['o'], g['pi'], steps=D, calc_mean=bool(g['calc_mean']), samples=S, stage=st, eps=eps_rollout(int(g['nseed']), M, D, S, st))
    upd('rollout sum_G', sG, g['sum_G']); upd('rollout t0', T[0], g['t0']); upd('rollout t1', T[1], g['t1'])
# large launches (the persistent split kernels need more than 128 images): 1 100 decoder rows against the oracle
from oracle import efe_oracle as EO
w = synth.make_weights(1234, 1.15)
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=31, init_weights=False)
m.load_flat_weights(w)
s_ = PX.uniform_fill(12, (1100, 10), 400, -1.5, 1.5)
got = m.model_down.decoder(s_, stage=5, pass_=PX.PASS_D2A)
with torch.no_grad():
    o = EO.OracleModel(w, EO.PhiloxNoise(31)).decoder(torch.from_numpy(s_[:300]), PX.PASS_D2A, 0, 5).numpy()
upd('decoder po, 1100-row launch vs oracle', got[:300], o)
print('engine option', OPT, '=', int(B3))
for k, v in worst.items():
    print(f'{k:40s} max |engine - reference| = {v:.3e}')
