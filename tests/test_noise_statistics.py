"""The PRODUCTION noise mode (device Philox) against the reference under ITS OWN generator.

Every other fixture is captured with torch's noise functions patched to the build's Philox stream, so oracle, injector and engine share
one keying scheme.  The `stats_*` fixtures (oracle/make_golden_stats.py) are SAMPLES of the unpatched reference
(/root/reference/src/torchmodel.py:270-300, 227-245, 354-393 under torch.manual_seed); here the engine's device-noise mode (-m gpu) and
the oracle with PhiloxNoise (CPU) must be statistically indistinguishable from them:

  mean            |m_a - m_b| <= Z * sqrt(v_a/n_a + v_b/n_b)
  variance        |ln(v_a / v_b)| <= Z * sqrt((k_a - 1)/n_a + (k_b - 1)/n_b)        k = fourth moment / variance^2 (the delta-method
                  standard error of ln s^2; reduces to the chi-square band 2/n for a normal sample)
  distribution    two-sample Kolmogorov-Smirnov, D <= c(alpha) * sqrt((n_a + n_b)/(n_a n_b)),  alpha = 1e-4 -> c = 2.225
  correlations    Fisher z of every pair of (term0, term1, term2_1, term2_2) and of G between the rows of one call:
                  |atanh r_a - atanh r_b| <= Z * sqrt(1/(n_a - 3) + 1/(n_b - 3))      <- what a keying collision would move
  actions         two-sample binomial bound on the frequency of every action at every simulation step; the first step also
                  against the habit posterior itself

Z = 4.5: with ~200 comparisons per test a correct build fails by chance with probability < 2e-3 -- and the engine's stream is
deterministic, so a green test stays green.  Beside it, a pure-key test: no two logical draws of a planner decision + a rollout
share a Philox counter (a collision is the one way two draws the reference makes independently could become correlated)."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import philox as PX
from oracle import synth
from oracle import efe_oracle as EO
from oracle import mcts_oracle as MO

Z = 4.5
KS_C = 2.225


# ---------------------------------------------------------------------------------------------------------------------------
# two-sample statistics
# ---------------------------------------------------------------------------------------------------------------------------
def _moments(x):
    x = np.asarray(x, dtype=np.float64)
    m = x.mean()
    d = x - m
    v = (d ** 2).mean()
    k = (d ** 4).mean() / max(v * v, 1e-300)
    return len(x), m, v * len(x) / (len(x) - 1), k


def ks_distance(a, b):
    a = np.sort(np.asarray(a, dtype=np.float64)); b = np.sort(np.asarray(b, dtype=np.float64))
    allv = np.concatenate([a, b])
    return float(np.max(np.abs(np.searchsorted(a, allv, side='right') / len(a) - np.searchsorted(b, allv, side='right') / len(b))))


def same_distribution(ref, x, name, report, ks=True):
    """asserts mean / variance / KS agreement of two 1-D samples; appends the observed statistics to `report`"""
    n1, m1, v1, k1 = _moments(ref)
    n2, m2, v2, k2 = _moments(x)
    z_mean = (m2 - m1) / math.sqrt(v1 / n1 + v2 / n2)
    z_var = math.log(v2 / v1) / math.sqrt(max(k1 - 1.0, 0.5) / n1 + max(k2 - 1.0, 0.5) / n2)
    d = ks_distance(ref, x) / math.sqrt((n1 + n2) / (n1 * n2)) if ks else 0.0
    report.append((name, z_mean, z_var, d))
    assert abs(z_mean) <= Z, f'{name}: mean {m2:.6g} vs reference {m1:.6g} = {z_mean:.2f} standard errors'
    assert abs(z_var) <= Z, f'{name}: variance {v2:.6g} vs reference {v1:.6g} = {z_var:.2f} standard errors of ln s^2'
    assert d <= KS_C, f'{name}: KS distance {d:.3f} (scaled) > {KS_C}'


def same_correlation(ref_a, ref_b, x_a, x_b, name, report):
    """Fisher z of corr(a, b) in both samples; columns (the action rows of a call) are compared one by one AND pooled (the mean of
    their z differences: four independent estimates, half the standard error)"""
    ref_a, ref_b, x_a, x_b = [np.asarray(v, np.float64).reshape(len(v), -1) for v in (ref_a, ref_b, x_a, x_b)]
    se = math.sqrt(1.0 / (len(ref_a) - 3) + 1.0 / (len(x_a) - 3))
    dz = []
    for j in range(ref_a.shape[1]):
        r1 = np.corrcoef(ref_a[:, j], ref_b[:, j])[0, 1]
        r2 = np.corrcoef(x_a[:, j], x_b[:, j])[0, 1]
        dz.append(math.atanh(r2) - math.atanh(r1))
        report.append((f'{name}[{j}] corr', r1, r2, dz[-1] / se))
        assert abs(dz[-1]) <= Z * se, f'{name}[{j}]: correlation {r2:.4f} vs reference {r1:.4f} = {dz[-1] / se:.2f} standard errors'
    zp = float(np.mean(dz)) / (se / math.sqrt(len(dz)))
    report.append((f'{name} pooled corr', 0.0, float(np.mean(dz)), zp))
    assert abs(zp) <= Z, f'{name}: pooled correlation difference {np.mean(dz):.4f} = {zp:.2f} standard errors'


def same_frequencies(ref_counts, n_ref, counts, n, name, report):
    for a, (c1, c2) in enumerate(zip(ref_counts, counts)):
        p = (c1 + c2) / (n_ref + n)
        se = math.sqrt(max(p * (1 - p), 1e-12) * (1.0 / n_ref + 1.0 / n))
        z = (c2 / n - c1 / n_ref) / se
        report.append((f'{name} action {a}', c1 / n_ref, c2 / n, z))
        assert abs(z) <= Z, f'{name}: action {a} frequency {c2 / n:.4f} vs reference {c1 / n_ref:.4f} = {z:.2f} standard errors'


def compare_calcG(ref, got, label, report):
    """ref / got: dicts of [N, 4] arrays G, t0, t1, t2, t2_1, t2_2 (row a of a call = action a)"""
    for key in ('G', 't0', 't1', 't2', 't2_1', 't2_2'):
        for a in range(4):
            same_distribution(ref[key][:, a], got[key][:, a], f'{label} {key}[{a}]', report)
    for ka, kb in (('t2_1', 't2_2'), ('t0', 't2_1'), ('t0', 't2_2'), ('t1', 't2_1'), ('t1', 't2_2'), ('t0', 't1')):
        same_correlation(ref[ka], ref[kb], got[ka], got[kb], f'{label} ({ka},{kb})', report)
    ab = [(0, 1), (1, 2), (2, 3), (0, 3), (0, 2), (1, 3)]            # rows of one call are independent draws in the reference
    for key in ('G', 't2_1', 't2_2'):
        same_correlation(ref[key][:, [a for a, _ in ab]], ref[key][:, [b for _, b in ab]], got[key][:, [a for a, _ in ab]],
                         got[key][:, [b for _, b in ab]], f'{label} {key} between rows', report)


def compare_simulate(ref, G, actions, qpi, label, report):
    same_distribution(ref['G'], G, f'{label} G', report)
    n_ref, n = len(ref['G']), len(G)
    depth = ref['actions'].shape[1]
    for t in range(depth):
        same_frequencies(np.bincount(ref['actions'][:, t], minlength=4), n_ref, np.bincount(actions[:, t], minlength=4), n, f'{label} step {t}', report)
    np.testing.assert_allclose(qpi, ref['Qpi'], rtol=1e-5, atol=1e-6)                      # the habit net has no dropout: deterministic
    f0 = np.bincount(actions[:, 0], minlength=4) / n                                       # first action ~ Qpi itself (torchmodel.py:363-364)
    for a in range(4):
        q = float(ref['Qpi'][a])
        assert abs(f0[a] - q) <= Z * math.sqrt(q * (1 - q) / n) + 1e-9, f'{label}: first action {a}: {f0[a]:.4f} vs Qpi {q:.4f}'
    # consecutive actions: the pair table (step 0, step 1) carries the dependence through the sampled state
    pair_ref = np.bincount(ref['actions'][:, 0] * 4 + ref['actions'][:, 1], minlength=16)
    pair = np.bincount(actions[:, 0] * 4 + actions[:, 1], minlength=16)
    same_frequencies(pair_ref, n_ref, pair, n, f'{label} steps (0,1) pair', report)


def _dump(report, capsys=None):
    worst = sorted(report, key=lambda r: -max(abs(v) for v in r[1:] if isinstance(v, float) and abs(v) < 1e6 and v == v))[:6]
    print('\n'.join(str(r) for r in worst))


# ---------------------------------------------------------------------------------------------------------------------------
# CPU: key uniqueness, and the oracle's Philox noise against the reference's distribution
# ---------------------------------------------------------------------------------------------------------------------------
class CounterLog:
    """records every Philox counter the oracle's noise draws (wraps oracle.philox.philox4x32_10)"""

    def __init__(self):
        self.blocks = []
        self._orig = PX.philox4x32_10

    def __enter__(self):
        def rec(c0, c1, c2, c3, k0, k1):
            b = np.broadcast_arrays(*[np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3)])
            lo = (b[0].reshape(-1) | (b[1].reshape(-1) << np.uint64(32)))
            hi = (b[2].reshape(-1) | (b[3].reshape(-1) << np.uint64(32)))
            self.blocks.append(np.stack([lo, hi], 1))
            return self._orig(c0, c1, c2, c3, k0, k1)
        PX.philox4x32_10 = rec
        return self

    def __exit__(self, *a):
        PX.philox4x32_10 = self._orig

    def counters(self):
        return np.concatenate(self.blocks, 0)


def test_no_two_logical_draws_share_a_philox_counter(weights_cache):
    """a planner decision of two lock-step episodes (root encode, expansions with S = 2, depth-3 simulations incl. the habit uniforms)
    followed by a rollout on the same model's stage counter: every dropout block, normal block and uniform is drawn at its own
    128-bit counter.  The oracle's draws ARE the engine's (bit-exact parity tests), so this is the engine's keying."""
    w = weights_cache(1234, 1.15)
    orc = EO.OracleModel(w, EO.PhiloxNoise(7))
    frames = synth.make_frames(23, 2)
    params = MO.Params(repeats=3, simulation_depth=3, use_means=False, samples=2, threshold=2.0, simulation_repeats=2)
    o = torch.from_numpy(np.repeat(synth.make_frames(21, 2), 4, axis=0))
    with CounterLog() as log, torch.no_grad():
        for e in range(2):                                   # lock-step episodes share the stages and differ in their rows
            MO.plan(orc, frames[e], params, stage0=100, episode=e)
        n_stages = 1 + (1 + params.repeats) + params.repeats * params.simulation_repeats
        orc.calculate_G_repeated(o, torch.eye(4).repeat(2, 1), 2, False, 2, 100 + n_stages)
        orc.calculate_G_4_repeated(o[:4], 2, True, 1, 100 + n_stages + 2)
    c = log.counters()
    assert len(c) > 40000
    uniq = np.unique(c, axis=0)
    assert len(uniq) == len(c), f'{len(c) - len(uniq)} Philox counters are drawn twice'
    # and the counter fields cannot alias: blk < 2^16 (tag above it), sample < 2^16 (pass above it)
    assert int((c[:, 0] & np.uint64(0xFFFF)).max()) < 16384 // 128 + 1
    assert int(((c[:, 1] & np.uint64(0xFFFFFFFF)) & np.uint64(0xFFFF)).max()) < 16


def _oracle_calcG_samples(orc, s0, S, n, stage):
    """n independent calculate_G draws as ONE oracle call over 4n rows (row 4k + a = replica k, action a: independent by global row)"""
    s0_t = torch.from_numpy(np.tile(s0, (n, 1)))
    pi = torch.eye(4).repeat(n, 1)
    with torch.no_grad():
        G, terms, _, _, _ = orc.calculate_G(s0_t, pi, S, stage)
    t21, t22 = orc.last_term2_parts
    return {'G': G.numpy().reshape(n, 4), 't0': terms[0].numpy().reshape(n, 4), 't1': terms[1].numpy().reshape(n, 4),
            't2': terms[2].numpy().reshape(n, 4), 't2_1': t21.numpy().reshape(n, 4), 't2_2': t22.numpy().reshape(n, 4)}


def test_oracle_philox_noise_has_the_reference_distribution(weights_cache):
    """the CPU half: the oracle with the build's Philox noise is statistically the unpatched reference (384 replicas: the GPU test
    runs the full 4096)"""
    ref = load_golden('stats_calcG')
    orc = EO.OracleModel(weights_cache(int(ref['wseed']), float(ref['gain'])), EO.PhiloxNoise(11))
    got = _oracle_calcG_samples(orc, ref['s0'], int(ref['samples']), 384, stage=3)
    report = []
    compare_calcG(ref, got, 'oracle calculate_G', report)
    _dump(report)


def test_oracle_simulate_has_the_reference_distribution(weights_cache):
    ref = load_golden('stats_simulate')
    orc = EO.OracleModel(weights_cache(int(ref['wseed']), float(ref['gain'])), EO.PhiloxNoise(11))
    n, depth = 320, int(ref['depth'])
    G = np.zeros(n, np.float32); A = np.zeros((n, depth), np.int64)
    start = torch.from_numpy(ref['start'])
    with torch.no_grad():
        for e in range(n):
            g, pi0, q = orc.mcts_step_simulate(start, depth, False, 5, episode=e)
            G[e] = g; A[e] = pi0.numpy().argmax(1)
    report = []
    compare_simulate(ref, G, A, q.numpy(), 'oracle simulate', report)
    _dump(report)


def _build_rocrand_check(tmp_path):
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / 'rocrand_philox_check')
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-I' + os.path.join(ROOT, 'deep-active-inference-mc_amd', 'csrc'),
           os.path.join(ROOT, 'tests', 'rocrand_philox_check.hip'), '-o', exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_oracle_philox_equals_the_rocrand_host_engine(tmp_path):
    """north_star names rocRAND: oracle/philox.py (= csrc/philox.h, bit-exact parity tests) produces the words of rocRAND's own
    rocrand_state_philox4x32_10 for the same key / counter -- here through rocRAND's HOST engine, 4096 counters, no GPU"""
    import subprocess
    exe = _build_rocrand_check(tmp_path)
    r = subprocess.run([exe, '--host'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    t = np.array([[int(v) for v in line.split()] for line in r.stdout.strip().splitlines()], dtype=np.uint64)
    assert t.shape == (4096, 10)
    for row in t:
        w = PX.philox4x32_10(row[0], row[1], row[2], row[3], int(row[4]), int(row[5]))
        assert [int(x) for x in w] == [int(x) for x in row[6:]], row


@pytest.mark.gpu
def test_engine_philox_equals_rocrand_on_the_device(tmp_path):
    """csrc/philox.h::noise_words == rocrand4(rocrand_init(seed, subsequence, offset)) on the GPU, 4096 key / counter pairs"""
    import subprocess
    exe = _build_rocrand_check(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'rocrand_philox_check OK' in r.stdout


# ---------------------------------------------------------------------------------------------------------------------------
# GPU: the engine's device-noise mode (what production runs) against the reference's distribution
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def model(weights_cache):
    import daimc_amd
    m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=2026, init_weights=False)
    m.load_flat_weights(weights_cache(1234, 1.15))
    assert m.eps_source is None and m.u_source is None           # device generator: normals, masks and uniforms all on the GPU
    return m


def _c(t):
    return t.detach().cpu().numpy()


def _engine_calcG(m, s0, pi, S, stage):
    parts = []
    G, terms, _, _, _ = m.calculate_G(s0, pi, samples=S, stage=stage, _parts=parts)
    return [_c(G), _c(terms[0]), _c(terms[1]), _c(terms[2]), _c(parts[0][0]), _c(parts[0][1])]


@pytest.mark.gpu
def test_device_noise_calculate_G_over_stages_has_the_reference_distribution(model):
    """4096 calls of calculate_G(s0 x 4, eye(4), samples = 3), one noise stage each -- the reference's own call shape"""
    ref = load_golden('stats_calcG')
    N, S = len(ref['G']), int(ref['samples'])
    out = np.zeros((6, N, 4), np.float32)
    s0 = torch.from_numpy(ref['s0']).cuda(); pi = torch.eye(4).cuda()
    for k in range(N):
        out[:, k] = _engine_calcG(model, s0, pi, S, stage=1000 + k)
    got = dict(zip(('G', 't0', 't1', 't2', 't2_1', 't2_2'), out))
    report = []
    compare_calcG(ref, got, 'engine calculate_G / stages', report)
    _dump(report)


@pytest.mark.gpu
def test_device_noise_calculate_G_over_rows_has_the_reference_distribution(model):
    """16 x the reference's 4096 draws as ONE call over 262 144 rows (independence by global row: what the batched planner and the
    multi-GPU sharding rely on; the standard errors are then the reference sample's alone), on two seeds"""
    ref = load_golden('stats_calcG')
    N, S = 16 * len(ref['G']), int(ref['samples'])
    s0 = torch.from_numpy(np.tile(ref['s0'], (N, 1))).cuda(); pi = torch.eye(4).repeat(N, 1).cuda()
    for seed in (2026, 0x9E3779B97F4A7C15):
        model.seed = seed
        out = _engine_calcG(model, s0, pi, S, stage=77)
        got = {k: v.reshape(N, 4) for k, v in zip(('G', 't0', 't1', 't2', 't2_1', 't2_2'), out)}
        report = []
        compare_calcG(ref, got, f'engine calculate_G / rows (seed {seed:#x})', report)
        _dump(report)
    model.seed = 2026


@pytest.mark.gpu
def test_device_noise_rollout_has_the_reference_distribution(model):
    """calculate_G_repeated(o x 4, eye(4), steps = 2, samples = 2): root encoder noise + the state carried between stages"""
    ref = load_golden('stats_rollout')
    N, D, S = len(ref['sum_G']), int(ref['steps']), int(ref['samples'])
    o = torch.from_numpy(np.tile(ref['o'], (N, 1, 1, 1))).cuda(); pi = torch.eye(4).repeat(N, 1).cuda()
    sum_G, terms, _ = model.calculate_G_repeated(o, pi, steps=D, samples=S, stage=5000)
    got = {'sum_G': _c(sum_G).reshape(N, 4), 't0': _c(terms[0]).reshape(N, 4), 't1': _c(terms[1]).reshape(N, 4), 't2': _c(terms[2]).reshape(N, 4)}
    # and the reference's own call shape: one 4-row call per draw, consecutive stages
    loop = np.zeros((N, 4), np.float32)
    o4 = torch.from_numpy(ref['o']).cuda(); pi4 = torch.eye(4).cuda()
    for k in range(N):
        loop[k] = _c(model.calculate_G_repeated(o4, pi4, steps=D, samples=S, stage=6000 + D * k)[0])
    report = []
    for key in ('sum_G', 't0', 't1', 't2'):
        for a in range(4):
            same_distribution(ref[key][:, a], got[key][:, a], f'engine rollout {key}[{a}]', report)
    for a in range(4):
        same_distribution(ref['sum_G'][:, a], loop[:, a], f'engine rollout / stages sum_G[{a}]', report)
    for ka, kb in (('t0', 't2'), ('t1', 't2'), ('t0', 't1')):
        same_correlation(ref[ka], ref[kb], got[ka], got[kb], f'engine rollout ({ka},{kb})', report)
    _dump(report)


@pytest.mark.gpu
def test_device_noise_simulate_has_the_reference_distribution(model):
    """mcts_step_simulate(start, depth 5): 2048 episodes of one simulate_batch call (k_sim_chain: device uniforms for the action
    draws), and 2048 one-episode calls over consecutive stages (the reference's call shape)"""
    ref = load_golden('stats_simulate')
    N, depth = len(ref['G']), int(ref['depth'])
    start = torch.from_numpy(np.tile(ref['start'], (N, 1))).cuda()
    G, pi0, q0 = model.simulate_batch(start, depth, False, stage=9000)
    report = []
    compare_simulate(ref, _c(G), _c(pi0).argmax(2), _c(q0)[0], 'engine simulate / episodes', report)
    Gs = np.zeros(N, np.float32); As = np.zeros((N, depth), np.int64)
    for k in range(N):
        g, p, q = model.mcts_step_simulate(ref['start'], depth, stage=9100 + k)
        Gs[k] = g; As[k] = _c(p).argmax(1)
    compare_simulate(ref, Gs, As, _c(q), 'engine simulate / stages', report)
    _dump(report)


@pytest.mark.gpu
@pytest.mark.parametrize('opt', ['mfma_bf16x3', 'mfma_f16x2'])
def test_device_noise_distributions_with_split_operands(model, opt):
    """the opt-in split-operand decoder kernels (csrc/bf16x3.hip: three bf16 planes / two fp16 planes per operand) under the SAME
    distribution tests, same bounds: calculate_G over 262 144 rows, the rollout over 4 096 rows and the 2 048-episode simulation are all
    launches of far more than 128 images, i.e. they run k_fc4_b3 / k_dec_a_b3 / k_dec_b_b3"""
    model.set_option(opt, 1)
    try:
        test_device_noise_calculate_G_over_rows_has_the_reference_distribution(model)
        test_device_noise_rollout_has_the_reference_distribution(model)
        test_device_noise_simulate_has_the_reference_distribution(model)
    finally:
        model.set_option(opt, 0)


@pytest.mark.gpu
def test_device_noise_planner_decisions_have_the_reference_distribution(model):
    """one level up: whole DECISIONS.  512 runs of the reference planner (active_inference_mcts, /root/reference/src/mcts.py:150-195, its
    default parameters at repeats 12 / depth 3 / threshold 0.3) under torch's own generator (oracle/make_golden_stats_planner.py) against 2 048 episodes of
    the lock-step planner on the same frame in device-noise mode: the distribution of the early stop (repeats_done), of the most visited
    root action, of the first action of the returned path, of the path length, and the mean root visit distribution"""
    import daimc_amd
    ref = load_golden('stats_planner')
    p = daimc_amd.MCTS_Params()                                      # the reference's defaults (use_means True, C 1.0) ...
    assert (p.use_means, p.C) == (bool(ref['use_means']), float(ref['C']))
    p.repeats, p.simulation_depth, p.threshold = int(ref['repeats']), int(ref['simulation_depth']), float(ref['threshold'])      # ... 12 / 3 / 0.3
    E, n_ref = 2048, len(ref['repeats_done'])
    frames = torch.from_numpy(np.tile(ref['frame'], (E, 1, 1, 1)))
    model._stage = 20000
    out, visits = daimc_amd.active_inference_mcts_batch(model, frames, p, o_shape=(1, 64, 64))
    report = []
    R = p.repeats
    reps = np.array([o_[1] for o_ in out]); plen = np.array([len(o_[0]) for o_ in out])
    first = np.array([o_[0][0] if len(o_[0]) else -1 for o_ in out])
    ref_first = np.where(ref['path_len'] > 0, ref['paths'][:, 0], -1)
    same_frequencies(np.bincount(ref['repeats_done'], minlength=R + 1), n_ref, np.bincount(reps, minlength=R + 1), E, 'planner repeats_done', report)
    same_frequencies(np.bincount(ref_first + 1, minlength=5), n_ref, np.bincount(first + 1, minlength=5), E, 'planner first action (0 = empty path)', report)
    same_frequencies(np.bincount(ref['path_len'], minlength=R + 3), n_ref, np.bincount(plen, minlength=R + 3), E, 'planner path length', report)
    v = visits.numpy()
    ref_v = ref['root_N'] / ref['root_N'].sum(1, keepdims=True)
    same_frequencies(np.bincount(ref_v.argmax(1), minlength=4), n_ref, np.bincount(v.argmax(1), minlength=4), E, 'planner most visited root action', report)
    for a in range(4):
        same_distribution(ref_v[:, a], v[:, a], f'planner root visit share[{a}]', report, ks=False)      # (a lattice-valued quantity: mean / variance only)
    assert len(np.unique(reps)) >= 3 and 0 < (reps < R).mean() < 1              # the early stop is exercised AND not universal
    _dump(report)
