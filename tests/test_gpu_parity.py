"""GPU parity tests (-m gpu): the HIP engine, called through the C ABI via the Python mirror, against
(1) the fixtures captured from the shimmed reference and (2) the CPU oracle on fresh seeded inputs.

Tolerances (fp32; observed maxima over the fixtures in profiles/r3_observed_errors.txt, tools/measure_errors.py):
network outputs rtol 1e-5 / atol 2e-6 (observed 2.0e-6), sigmoid images atol 4e-6 (observed 1.8e-6); term0 / term1 atol 1e-4
(observed 1.1e-5 / 2.4e-6); term2 and G: atol = 1e-6 * max(|term2_1|, 1) + 5e-4, i.e. 3.3e-3 at |term2_1| = 2.8e3 (observed
7.3e-4 = 3 ulp of the 4096-pixel sums whose cancelling difference term2 is: the oracle's own fp32 rounding bounds it; 4.9e-4 in
round 2, before the gather's sigmoid / log terms moved to the hardware exp2 / rcp / log2 instructions)."""
import os

import numpy as np
import pytest
import torch

from conftest import eps_calcG, eps_rollout, usable_cores, load_golden
from oracle import philox as PX
from oracle import synth
from oracle import efe_oracle as EO

pytestmark = pytest.mark.gpu
GAINS = ['g100', 'g115', 'g135']


@pytest.fixture(scope='module')
def models(weights_cache):
    import daimc_amd
    cache = {}

    def get(wseed, gain, seed):
        key = (int(wseed), float(gain))
        if key not in cache:
            m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=int(seed), init_weights=False)
            m.load_flat_weights(weights_cache(wseed, gain))
            cache[key] = m
        m = cache[key]
        m.seed = int(seed)
        m.row_offset = 0
        m.eps_source, m.u_source = None, None
        return m
    return get


def inject(m):
    """injected-noise mode for whole planners: every call builds its normals / uniforms from the numpy Philox mirror, so both
    sides consume bit-identical noise (the device generator's Box-Muller differs from numpy's libm by an ulp)"""
    m.eps_source, m.u_source = PX.normals, PX.uniforms
    return m


def _model(g, models):
    return models(g['wseed'], g['gain'], g['nseed'])


def c(t):
    return t.detach().cpu().numpy()


def gtol(t21):
    return 1e-6 * max(float(np.max(np.abs(t21))), 1.0) + 5e-4


@pytest.mark.parametrize('gain', GAINS)
def test_networks_vs_golden(golden, models, gain):
    g = golden(f'nets_{gain}')
    m = _model(g, models)
    st, seed, M = int(g['stage']), int(g['nseed']), len(g['s'])
    eps_t = PX.normals(seed, M, 10, PX.PASS_T1, 0, st)
    ps1, mean, lv = m.model_mid.transition_with_sample(g['pi'], g['s'], stage=st, pass_=PX.PASS_T1, eps=eps_t)
    np.testing.assert_allclose(c(mean), g['t_mean'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(lv), g['t_lv'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(ps1), g['t_ps1'], rtol=1e-5, atol=2e-6)
    # device-generated normals (Box-Muller on Philox) agree with the numpy mirror to fp32 libm accuracy
    ps1_dev, _, _ = m.model_mid.transition_with_sample(g['pi'], g['s'], stage=st, pass_=PX.PASS_T1)
    np.testing.assert_allclose(c(ps1_dev), g['t_ps1'], rtol=1e-4, atol=1e-5)

    po = m.model_down.decoder(g['s'], stage=st, pass_=PX.PASS_D1)
    assert po.shape == (M, 1, 64, 64)
    np.testing.assert_allclose(c(po), g['d_po'], rtol=1e-5, atol=4e-6)

    eps_e = PX.normals(seed, M, 10, PX.PASS_E1, 0, st)
    s, emean, elv = m.model_down.encoder_with_sample(g['frames'], stage=st, pass_=PX.PASS_E1, eps=eps_e)
    np.testing.assert_allclose(c(emean), g['e_mean'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(elv), g['e_lv'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(s), g['e_s'], rtol=1e-5, atol=2e-6)

    logits, q, logq = m.model_top.encode_s(g['s'])
    np.testing.assert_allclose(c(logits), g['h_logits'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(q), g['h_q'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c(logq), g['h_logq'], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize('gain', GAINS)
@pytest.mark.parametrize('case', ['m4s1', 'm6s3'])
def test_calculate_G_vs_golden(golden, models, gain, case):
    g = golden(f'calcG_{case}_{gain}')
    m = _model(g, models)
    S, st, M = int(g['samples']), int(g['stage']), len(g['s0'])
    eps = eps_calcG(int(g['nseed']), M, S, st)
    parts = []
    G, terms, ps1, ps1_mean, po1 = m.calculate_G(g['s0'], g['pi0'], samples=S, stage=st, eps=eps, _parts=parts)
    np.testing.assert_allclose(c(ps1), g['ps1'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(ps1_mean), g['ps1_mean'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(po1), g['po1'], rtol=1e-5, atol=4e-6)
    np.testing.assert_allclose(c(terms[0]), g['t0'], atol=1e-4)
    np.testing.assert_allclose(c(terms[1]), g['t1'], atol=1e-4)
    np.testing.assert_allclose(c(parts[0][0]), g['t2_1'], rtol=1e-6, atol=1e-3)
    np.testing.assert_allclose(c(parts[0][1]), g['t2_2'], rtol=1e-6, atol=1e-3)
    np.testing.assert_allclose(c(terms[2]), g['t2'], atol=gtol(g['t2_1']))
    np.testing.assert_allclose(c(G), g['G'], atol=gtol(g['t2_1']))


@pytest.mark.parametrize('gain', GAINS)
def test_calculate_G_mean_vs_golden(golden, models, gain):
    g = golden(f'calcGmean_{gain}')
    m = _model(g, models)
    st = int(g['stage'])
    eps = eps_calcG(int(g['nseed']), 4, 1, st)
    G, terms, ps1_mean, po1 = m.calculate_G_mean(g['s0'], m.pi_one_hot, stage=st, eps=eps)
    np.testing.assert_allclose(c(ps1_mean), g['ps1_mean'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(po1), g['po1'], rtol=1e-5, atol=4e-6)
    np.testing.assert_allclose(c(terms[0]), g['t0'], atol=1e-4)
    np.testing.assert_allclose(c(terms[1]), g['t1'], atol=1e-4)
    np.testing.assert_allclose(c(G), g['G'], atol=gtol(g['t2_1']))


@pytest.mark.parametrize('name', ['rollout_cfg1', 'rollout_m8d2s2', 'rollout_m8d2s2mean'])
def test_rollout_vs_golden(golden, models, name):
    g = golden(name)
    m = _model(g, models)
    D, S, st, M = int(g['steps']), int(g['samples']), int(g['stage']), len(g['o'])
    eps = eps_rollout(int(g['nseed']), M, D, S, st)
    sum_G, terms, po1 = m.calculate_G_repeated(g['o'], g['pi'], steps=D, calc_mean=bool(g['calc_mean']), samples=S, stage=st, eps=eps)
    tol = D * gtol(np.array([2800.0]))
    np.testing.assert_allclose(c(terms[0]), g['t0'], atol=2e-4)
    np.testing.assert_allclose(c(terms[1]), g['t1'], atol=2e-4)
    np.testing.assert_allclose(c(terms[2]), g['t2'], atol=tol)
    np.testing.assert_allclose(c(sum_G), g['sum_G'], atol=tol)
    np.testing.assert_allclose(c(po1), g['po1'], rtol=1e-5, atol=2e-5)
    P, logP = m.action_posterior(sum_G)
    # posterior from OUR sum_G vs the reference's posterior from ITS sum_G (temperature-10 softmax, util.py:46-53)
    np.testing.assert_allclose(c(P), g['Ppi'], atol=5e-3)
    P2, logP2 = m.action_posterior(torch.from_numpy(g['sum_G']))
    np.testing.assert_allclose(c(P2), g['Ppi'], rtol=1e-5)
    np.testing.assert_allclose(c(logP2), g['logPpi'], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('name', ['rollout4_s2', 'rollout4_mean'])
def test_rollout4_vs_golden(golden, models, name):
    g = golden(name)
    m = _model(g, models)
    D, S, st = int(g['steps']), int(g['samples']), int(g['stage'])
    Seff = 1 if bool(g['calc_mean']) else S
    eps = eps_rollout(int(g['nseed']), 4, D, Seff, st)
    sum_G, terms, po1 = m.calculate_G_4_repeated(g['o'], steps=D, calc_mean=bool(g['calc_mean']), samples=S, stage=st, eps=eps)
    tol = D * gtol(np.array([2800.0]))
    np.testing.assert_allclose(c(terms[0]), g['t0'], atol=2e-4)
    np.testing.assert_allclose(c(terms[1]), g['t1'], atol=2e-4)
    np.testing.assert_allclose(c(sum_G), g['sum_G'], atol=tol)
    np.testing.assert_allclose(c(po1), g['po1'], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('name', ['simulate_sample', 'simulate_means'])
def test_simulate_vs_golden(golden, models, name):
    g = golden(name)
    m = _model(g, models)
    seed, st, ep, T = int(g['nseed']), int(g['stage']), int(g['episode']), int(g['depth'])
    # injected normals / uniforms in the layout of efe_simulate: step transitions, then the trajectory's (T1 unused, T2, D2B)
    eps = np.concatenate([PX.normals(seed, 1, 10, PX.PASS_SIM, t, st, ep).reshape(-1) for t in range(T)]
                         + [eps_calcG(seed, T, 1, st, ep * T).reshape(-1)])
    u = np.stack([PX.uniforms(seed, 1, PX.PASS_HABIT, t, st, ep) for t in range(T)])
    G, pi0, q = m.mcts_step_simulate(g['start'], T, use_means=bool(g['use_means']), stage=st, row_offset=ep, eps=eps, u=u)
    assert np.array_equal(c(pi0), g['pi0'])                 # same actions sampled (Philox uniform + habit posterior)
    np.testing.assert_allclose(c(q), g['Qpi'], rtol=1e-5, atol=1e-6)
    assert abs(G - float(g['G'])) < gtol(np.array([2800.0]))
    # device-generated noise (Philox + Box-Muller on the GPU): same actions, G within the libm difference of the normals
    G2, pi02, _ = m.mcts_step_simulate(g['start'], T, use_means=bool(g['use_means']), stage=st, row_offset=ep)
    assert np.array_equal(c(pi02), g['pi0']) and abs(G2 - float(g['G'])) < 5e-3


@pytest.mark.parametrize('name', ['simulate_invalid_nan', 'simulate_invalid_inf'])
def test_simulate_invalid_habit_posterior_vs_reference(golden, weights_cache, name):
    """the reference's bare-except fallback (/root/reference/src/torchmodel.py:362-367, 378-381) on the device: a habit network whose
    posterior torch.multinomial rejects (NaN everywhere / [0, 0, NaN, 0]) -> k_sim_chain takes action 0 on EVERY step and returns that
    one-hot as Qpi; fixtures captured from the reference with the poisoned output bias (oracle/make_golden_invalid.py).  Checked through
    the single-episode API, through the batched call beside healthy neighbours' rows, and in device-noise mode."""
    import daimc_amd
    g = golden(name)
    w = dict(weights_cache(int(g['wseed']), float(g['gain'])))
    b = np.array(w['top.qpi_net.4.bias'], copy=True)
    b[int(g['bias_index'])] = g['bias_value']
    w['top.qpi_net.4.bias'] = b
    seed, st, ep, T = int(g['nseed']), int(g['stage']), int(g['episode']), int(g['depth'])
    m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=seed, init_weights=False)
    m.load_flat_weights(w)
    eps = np.concatenate([PX.normals(seed, 1, 10, PX.PASS_SIM, t, st, ep).reshape(-1) for t in range(T)]
                         + [eps_calcG(seed, T, 1, st, ep * T).reshape(-1)])
    u = np.stack([PX.uniforms(seed, 1, PX.PASS_HABIT, t, st, ep) for t in range(T)])
    G, pi0, q = m.mcts_step_simulate(g['start'], T, use_means=False, stage=st, row_offset=ep, eps=eps, u=u)
    onehot0 = np.eye(4, dtype=np.float32)[[0] * T]
    assert np.array_equal(c(pi0), g['pi0']) and np.array_equal(c(pi0), onehot0)
    assert np.array_equal(c(q), g['Qpi']) and np.array_equal(c(q), [1, 0, 0, 0])
    assert abs(G - float(g['G'])) < gtol(np.array([2800.0]))
    # device-generated noise: the fallback does not consume the uniform, the normals differ by Box-Muller's libm
    G2, pi02, q2 = m.mcts_step_simulate(g['start'], T, use_means=False, stage=st, row_offset=ep)
    assert np.array_equal(c(pi02), onehot0) and np.array_equal(c(q2), [1, 0, 0, 0]) and abs(G2 - float(g['G'])) < 5e-3
    # the batched call (the lock-step planner's form): episode `ep` of a batch of ep + 2 starts at global rows
    starts = np.tile(np.asarray(g['start'], dtype=np.float32)[None], (ep + 2, 1))
    Gb, pib, qb = m.simulate_batch(starts, T, use_means=False, stage=st, row_offset=0)
    assert np.array_equal(c(pib).reshape(ep + 2, T, 4), np.tile(onehot0[None], (ep + 2, 1, 1)))
    assert np.array_equal(c(qb), np.tile(np.array([[1, 0, 0, 0]], dtype=np.float32), (ep + 2, 1)))
    assert abs(float(c(Gb)[ep]) - float(g['G'])) < 5e-3 and np.isfinite(c(Gb)).all()
    # a whole planner on this model: the prior is the one-hot at every node, the planner finishes and its visit counts are a distribution
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.use_means, p.threshold, p.samples = 6, 3, False, 2.0, 2
    out, visits = daimc_amd.active_inference_mcts_batch(m, torch.from_numpy(synth.make_frames(5, 2)[:, 0][:, None]), p, o_shape=(1, 64, 64))
    np.testing.assert_allclose(visits.sum(1).numpy(), 1.0, rtol=1e-6)
    assert all(np.isfinite(o_[4]).all() for o_ in out)


def test_lockstep_batch_at_benchmark_size_contains_the_reference_episodes(golden, models):
    """BASELINE configs[2] at its benchmarked batch: 64 episodes planned in lock-step (256-row x 10-sample expansions, simulations
    forked onto the second stream through the replica context, node capacity 1 + A (repeats + 2) at E = 64).  The noise keys are
    global, so episodes 0-2 of the batch -- the three frames of mcts_deep_s10 -- must reproduce the reference planner's capture
    whatever the other 61 episodes are; every episode's root visit distribution is a distribution."""
    import daimc_amd
    g = golden('mcts_deep_s10')
    m = inject(_model(g, models))
    p = _deep_params(g)
    E, E0 = 64, int(g['episodes'])
    frames = torch.cat([torch.from_numpy(g['frames']), torch.from_numpy(synth.make_frames(77, E - E0))], 0)
    m._stage = int(g['stage'])
    out, visits = daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
    planner = next(pl for pl in m._planners.values() if pl.E == E)
    assert planner.overlap and planner.sim_model is not m          # the second stream / replica context was in use
    assert planner.cap == 1 + 4 * (p.repeats + 2)
    for e in range(E0):
        _check_deep(g, e, out[e], visits[e])
    np.testing.assert_allclose(visits.sum(1).numpy(), 1.0, rtol=1e-6)
    assert all(len(o_[3]) == o_[1] == p.repeats for o_ in out)       # threshold 2.0 in the fixture: every iteration of every episode ran
    assert int(planner.n_nodes.max()) <= planner.cap


@pytest.mark.parametrize('name', ['mcts_deep_s10_thr', 'mcts_deep_s10_thr04'])
def test_lockstep_batch_with_early_stops_contains_the_reference_episodes(golden, models, name):
    """the path bench.py's threshold-0.5 leg runs, against the reference planner's own early stops at the benchmark's depth
    (oracle/make_golden_thr.py: /root/reference/src/mcts.py:170-181 at threshold 0.5 -> episodes stop before iterations 41, 50, 50, 50, 29,
    50; at 0.4 -> 28, 45, 19, 50, 19, 42): 64 episodes in lock-step, the lagged host check, stopped episodes compacted out of the batch
    (efe_rows.ids) while the others keep planning -- episodes 0-5 must reproduce the capture: repeats_done, every path, the G history,
    the final path, the root visit distribution at the moment of the stop."""
    import daimc_amd
    g = golden(name)
    m = inject(_model(g, models))
    p = _deep_params(g)
    E, E0 = 64, int(g['episodes'])
    frames = torch.cat([torch.from_numpy(g['frames']), torch.from_numpy(synth.make_frames(78, E - E0))], 0)
    m._stage = int(g['stage'])
    out, visits = daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
    planner = next(pl for pl in m._planners.values() if pl.E == E and pl.p.threshold == p.threshold)
    assert planner.overlap and planner._ids is not None and len(planner._ids[1]) < E          # the batch WAS compacted
    for e in range(E0):
        _check_deep(g, e, out[e], visits[e])
    stops = [o_[1] for o_ in out]
    assert min(stops) < p.repeats and max(stops) == p.repeats           # some of the other 58 stop early too, some run to the end
    np.testing.assert_allclose(visits.sum(1).numpy(), 1.0, rtol=1e-6)
    # the same batch without compaction and without the lagged check (masked only, host reads the count every 8th iteration): identical
    q = _deep_params(g)
    q.compact_stopped, q.lagged_check = False, False
    m._stage = int(g['stage'])
    out2, visits2 = daimc_amd.active_inference_mcts_batch(m, frames, q, o_shape=(1, 64, 64))
    assert [(o_[0], o_[1], o_[2], o_[3]) for o_ in out] == [(o_[0], o_[1], o_[2], o_[3]) for o_ in out2]
    assert all(o_[4] == o2[4] for o_, o2 in zip(out, out2)) and torch.equal(visits, visits2)


def test_unfused_transition_option_covers_every_path(models):
    """A/B option mid_unfused (layer-by-layer transition MLP through k_dense instead of k_trans_fused): calculate_G AND the trajectory core
    behind mcts_step_simulate (whose loop-2 transition otherwise comes from k_sim_chain) follow it -- same masks and normals, fp32
    summation order of the 512-wide layers differs"""
    m = models(1234, 1.15, 29)
    s0 = PX.uniform_fill(6, (8, 10), 310, -1, 1)
    starts = PX.uniform_fill(6, (5, 10), 311, -1, 1)
    ref_G = c(m.calculate_G(s0, torch.eye(4).repeat(2, 1), samples=2, stage=3)[0])
    ref_sim = [c(t) for t in m.simulate_batch(starts, 4, False, stage=9)]
    try:
        m.set_option('mid_unfused', 1)
        G = c(m.calculate_G(s0, torch.eye(4).repeat(2, 1), samples=2, stage=3)[0])
        sim = [c(t) for t in m.simulate_batch(starts, 4, False, stage=9)]
    finally:
        m.set_option('mid_unfused', 0)
    np.testing.assert_allclose(G, ref_G, atol=gtol(np.array([2800.0])))
    np.testing.assert_allclose(sim[0], ref_sim[0], atol=gtol(np.array([2800.0])))
    assert np.array_equal(sim[1], ref_sim[1])                 # the sampled actions (k_sim_chain's own rollout is the same launch either way)
    np.testing.assert_allclose(sim[2], ref_sim[2], rtol=1e-6)
    assert not np.array_equal(sim[0], ref_sim[0]) or not np.array_equal(G, ref_G)      # the option really switched the kernels


@pytest.mark.parametrize('E', [1, 3, 8, 11, 16])
def test_split_simulation_chain_is_bit_identical(models, E):
    """efe_simulate of <= 16 episodes runs the habit-policy chain on EIGHT workgroups per 8 episodes (k_sim_chain<8>: the two 512-wide
    transition layers split by feature tiles, slices exchanged with agent-scope accesses and a counter) -- every output bit equals the
    one-workgroup kernel's (option sim_split = 0), call after call (the counters re-arm themselves), with device and with injected noise"""
    m = models(1234, 1.15, 37)
    starts = PX.uniform_fill(13, (E, 10), 610, -1, 1)
    outs = {}
    try:
        for split in (1, 0):
            m.set_option('sim_split', split)
            runs = []
            for rep in range(3):
                runs.append([c(t) for t in m.simulate_batch(starts, 5, False, stage=40 + rep)])
            inject(m)
            runs.append([c(t) for t in m.simulate_batch(starts, 4, True, stage=50)])
            m.eps_source, m.u_source = None, None
            outs[split] = runs
    finally:
        m.set_option('sim_split', 1)
    for ra, rb in zip(outs[1], outs[0]):
        for a_, b_ in zip(ra, rb):
            assert np.isfinite(a_).all() and np.array_equal(a_, b_)
    assert not np.array_equal(outs[1][0][0], outs[1][1][0])          # different stages: different draws


def test_planner_replica_follows_engine_options(models):
    """engine options are per context: the replica the lock-step planner simulates on must compute with the options of the model it
    mirrors (reward_upstream_intent changes term0 / G) -- a planner with the simulations on the second stream equals the same planner
    with them on the main context, with the option set BEFORE and AFTER the replica exists"""
    import daimc_amd
    m = models(1234, 1.15, 23)
    frames = torch.from_numpy(synth.make_frames(58, 8)[:, 0][:, None])

    def plan(overlap):
        p = daimc_amd.MCTS_Params()
        p.repeats, p.simulation_depth, p.use_means, p.threshold, p.samples = 8, 3, False, 2.0, 2
        p.overlap_simulate = overlap
        m._stage = 300
        out, visits = daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
        return [o_[3] for o_ in out], np.array([o_[4] for o_ in out]), visits
    try:
        base = plan(True)                              # creates the replica with default options
        m.set_option('reward_upstream_intent', 1)      # ... which must follow
        a_paths, a_G, a_v = plan(True)
        b_paths, b_G, b_v = plan(False)
        assert a_paths == b_paths and np.array_equal(a_G, b_G) and torch.equal(a_v, b_v)
        assert not np.array_equal(a_G, base[1])        # the option does change G
        m.__dict__.pop('_replica', None)               # a replica created AFTER the option was set
        c_paths, c_G, c_v = plan(True)
        assert c_paths == b_paths and np.array_equal(c_G, b_G)
    finally:
        m.set_option('reward_upstream_intent', 0)


@pytest.mark.parametrize('name', ['mcts_means', 'mcts_samples', 'mcts_prior'])
def test_mcts_vs_golden(golden, models, name):
    import daimc_amd
    g = golden(name)
    m = inject(_model(g, models))
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.use_means, p.threshold = int(g['repeats']), int(g['simulation_depth']), bool(g['use_means']), float(g['threshold'])
    p.using_prior_for_exploration = (name == 'mcts_prior')
    for host_tree in (False, True):          # the device-resident planner (default) and the host-side Node tree of the reference's API
        p.host_tree = host_tree
        m._stage = int(g['stage'])
        path, reps, explored, all_paths, all_G = daimc_amd.active_inference_mcts(m, torch.from_numpy(g['frame']), p, o_shape=(1, 64, 64))
        assert reps == int(g['repeats_done']) and explored == int(g['states_explored'])
        ref_paths = [[int(a) for a in row if a >= 0] for row in g['all_paths']]
        assert [[int(a) for a in pth] for pth in all_paths] == ref_paths
        np.testing.assert_allclose(np.array(all_G), g['all_paths_G'], atol=gtol(np.array([2800.0])))
        assert [int(a) for a in path] == [int(a) for a in g['final_path']]


# ------------------------------------------------------------------------------------------------------
# fresh seeded inputs vs the oracle (sizes the oracle finishes in seconds), ragged / edge cases
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,S', [(1, 1), (3, 2), (33, 1), (70, 2)])
def test_calculate_G_vs_oracle_ragged(models, weights_cache, M, S):
    seed, st, ro = 99, 5, 1000
    w = weights_cache(1234, 1.15)
    m = models(1234, 1.15, seed)
    orc = EO.OracleModel(w, EO.PhiloxNoise(seed, row_offset=ro))
    s0 = PX.uniform_fill(8, (M, 10), 300 + M, -1.0, 1.0)
    pi0 = np.eye(4, dtype=np.float32)[np.arange(M) % 4]
    with torch.no_grad():
        oG, oT, ops1, ops1m, opo1 = orc.calculate_G(torch.from_numpy(s0), torch.from_numpy(pi0), S, st)
    eps = eps_calcG(seed, M, S, st, ro)
    G, terms, ps1, ps1m, po1 = m.calculate_G(s0, pi0, samples=S, stage=st, eps=eps, row_offset=ro)
    np.testing.assert_allclose(c(ps1), ops1.numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(po1), opo1.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c(G), oG.numpy(), atol=gtol(orc.last_term2_parts[0].numpy()))


def test_row_offset_invariance(models):
    """rows keyed globally: evaluating rows [0,8) at once == evaluating [0,4) and [4,8) separately
    (what makes 1/2/4/8-GPU results identical)."""
    m = models(1234, 1.15, 11)
    o = synth.make_frames(31, 8)
    pi = np.eye(4, dtype=np.float32)[np.arange(8) % 4]
    G_all, _, _ = m.calculate_G_repeated(o, pi, steps=2, samples=2, stage=0, row_offset=0)
    G_a, _, _ = m.calculate_G_repeated(o[:4], pi[:4], steps=2, samples=2, stage=0, row_offset=0)
    G_b, _, _ = m.calculate_G_repeated(o[4:], pi[4:], steps=2, samples=2, stage=0, row_offset=4)
    assert torch.equal(G_all, torch.cat([G_a, G_b]))


def test_chunking_invariance(models):
    m = models(1234, 1.15, 12)
    o = synth.make_frames(32, 6)
    pi = np.eye(4, dtype=np.float32)[np.arange(6) % 4]
    G1, _, po1 = m.calculate_G_repeated(o, pi, steps=2, samples=3, stage=3)
    m.set_option('dec_chunk', 7); m.set_option('enc_chunk', 5)
    G2, _, po2 = m.calculate_G_repeated(o, pi, steps=2, samples=3, stage=3)
    m.set_option('dec_chunk', 32768); m.set_option('enc_chunk', 32768)
    assert torch.equal(G1, G2) and torch.equal(po1, po2)


def test_no_experiment_switches_in_the_product_library(models):
    """the wrong-result timing switches are patches under tools/ubench/patches/ (tools/ubench/build_alt.py): the shipped library has no
    option that reaches them"""
    m = models(1234, 1.15, 13)
    for opt in ('dbg_a', 'dbg_b', 'tl_buf'):
        with pytest.raises(RuntimeError, match='unknown option'):
            m.set_option(opt, 1)


def test_reward_upstream_intent_option(models, weights_cache):
    """the reward target the upstream NHWC code means (top three rows, left half = 1) as an engine option beside the
    replicated NCHW-broadcast quirk (default).  check_reward and the fused decoder epilogue against the oracle restatement
    (oracle.efe_oracle.check_reward_upstream_intent); switching the option back restores the pinned values bit for bit."""
    seed, st, M, S = 17, 3, 6, 2
    w = weights_cache(1234, 1.15)
    m = models(1234, 1.15, seed)
    orc = EO.OracleModel(w, EO.PhiloxNoise(seed))
    s0 = PX.uniform_fill(4, (M, 10), 77, -1.0, 1.0)
    pi0 = np.eye(4, dtype=np.float32)[np.arange(M) % 4]
    eps = eps_calcG(seed, M, S, st)
    G0, T0, _, _, po0 = m.calculate_G(s0, pi0, samples=S, stage=st, eps=eps)
    m.set_option('reward_upstream_intent', 1)
    try:
        orc.reward_upstream_intent = True
        with torch.no_grad():
            oG, oT, _, _, opo1 = orc.calculate_G(torch.from_numpy(s0), torch.from_numpy(pi0), S, st)
            ocr = orc.check_reward(opo1)
        G1, T1, _, _, po1 = m.calculate_G(s0, pi0, samples=S, stage=st, eps=eps)
        assert torch.equal(po1, po0)                                        # the images do not depend on the reward target
        np.testing.assert_allclose(c(T1[0]), oT[0].numpy(), atol=1e-4)
        np.testing.assert_allclose(c(T1[1]), oT[1].numpy(), atol=1e-4)
        np.testing.assert_allclose(c(G1), oG.numpy(), atol=gtol(orc.last_term2_parts[0].numpy()))
        np.testing.assert_allclose(c(m.check_reward(opo1.numpy())), ocr.numpy(), rtol=2e-6, atol=1e-6)
        gi = load_golden('helpers_intent')          # the reference's formula on the NHWC view of a batch (oracle/make_golden_deep.py)
        np.testing.assert_allclose(c(m.check_reward(gi['p'])), gi['reward_upstream_intent'], rtol=2e-6, atol=1e-6)
        assert not np.allclose(c(T1[0]), c(T0[0]), atol=1e-2)              # a different quantity than the replicated quirk
    finally:
        m.set_option('reward_upstream_intent', 0)
    G2, T2, _, _, _ = m.calculate_G(s0, pi0, samples=S, stage=st, eps=eps)
    assert torch.equal(G2, G0) and torch.equal(T2[0], T0[0])


def test_full_size_properties(models):
    """BASELINE cfg-2 shape (128 rows, S=10, D=5): size-independent properties -- determinism, finite
    outputs, duplicate rows with equal global ids give equal results, action posterior sums to one."""
    m = models(1234, 1.15, 1)
    fr = synth.make_frames(41, 32)
    o = np.repeat(fr, 4, axis=0)
    pi = np.tile(np.eye(4, dtype=np.float32), (32, 1))
    G1, t1, _ = m.calculate_G_repeated(o, pi, steps=5, samples=10, stage=0)
    G2, t2, _ = m.calculate_G_repeated(o, pi, steps=5, samples=10, stage=0)
    assert torch.equal(G1, G2)
    assert torch.isfinite(G1).all()
    np.testing.assert_allclose(c(-t1[0] + t1[1] + t1[2]), c(G1), rtol=1e-5, atol=1e-2)
    P, logP = m.action_posterior(G1)
    np.testing.assert_allclose(c(P.sum(1)), 1.0, rtol=1e-5)
    # a different stage gives different MC noise, but a statistically similar G
    G3, _, _ = m.calculate_G_repeated(o, pi, steps=5, samples=10, stage=100)
    assert not torch.equal(G1, G3)


def test_device_noise_mode_vs_oracle(models, weights_cache):
    """production noise mode (masks AND normals generated on the device from Philox, nothing injected) against the oracle
    consuming the same Philox stream through the numpy mirror: the only difference is Box-Muller's libm (an ulp on the
    normals), so G agrees to ~1e-3 per stage; checked over 6 noise stages, which also shows the MC spread between stages
    is real (stages differ by far more than the tolerance)."""
    seed, S = 5, 4
    w = weights_cache(1234, 1.0)
    m = models(1234, 1.0, seed)
    orc = EO.OracleModel(w, EO.PhiloxNoise(seed))
    s0 = np.tile(PX.uniform_fill(9, (1, 10), 77, -1, 1), (4, 1))
    gs = []
    for st in range(6):
        G, _, _, _, _ = m.calculate_G(s0, m.pi_one_hot, samples=S, stage=st)
        with torch.no_grad():
            oG = orc.calculate_G(torch.from_numpy(s0), torch.eye(4), S, st)[0]
        np.testing.assert_allclose(c(G), oG.numpy(), atol=gtol(orc.last_term2_parts[0].numpy()) + 4e-3)
        gs.append(c(G))
    gs = np.stack(gs)
    assert (gs.std(0) > 0.05).all()          # different stages = different MC noise


@pytest.mark.parametrize('name', ['mcts_means', 'mcts_samples', 'mcts_prior'])
def test_batched_mcts_single_episode_equals_golden(golden, models, name):
    """the lock-step planner with E = 1 reproduces the reference's single-episode decision"""
    import daimc_amd
    g = golden(name)
    m = inject(_model(g, models))
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.use_means, p.threshold = int(g['repeats']), int(g['simulation_depth']), bool(g['use_means']), float(g['threshold'])
    p.using_prior_for_exploration = (name == 'mcts_prior')
    m._stage = int(g['stage'])
    out, dist = daimc_amd.active_inference_mcts_batch(m, torch.from_numpy(g['frame'])[None], p, o_shape=(1, 64, 64))
    path, reps, explored, all_paths, all_G = out[0]
    assert reps == int(g['repeats_done']) and explored == int(g['states_explored'])
    assert all_paths == [[int(a) for a in row if a >= 0] for row in g['all_paths']]
    np.testing.assert_allclose(np.array(all_G), g['all_paths_G'], atol=gtol(np.array([2800.0])))
    assert [int(a) for a in path] == [int(a) for a in g['final_path']]
    np.testing.assert_allclose(dist.sum(1).numpy(), 1.0, rtol=1e-6)


def _deep_params(g):
    import daimc_amd
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.use_means, p.threshold = int(g['repeats']), int(g['simulation_depth']), False, float(g['threshold'])
    p.samples = int(g['samples'])
    p.using_prior_for_exploration, p.use_habit = bool(g['using_prior_for_exploration']), bool(g['use_habit'])
    return p


def _check_deep(g, e, res, visits=None):
    path, reps, explored, all_paths, all_G = res
    n = int(g['n_paths'][e])
    assert reps == int(g['repeats_done'][e]) and explored == int(g['states_explored'][e]) and len(all_paths) == n
    assert all_paths == [[int(a) for a in row if a >= 0] for row in g['all_paths'][e][:n]]
    np.testing.assert_allclose(np.array(all_G), g['all_paths_G'][e][:n], atol=gtol(np.array([2800.0])))
    assert [int(a) for a in path] == [int(a) for a in g['final_path'][e] if a >= 0]
    if visits is not None:
        np.testing.assert_array_equal(visits.numpy(), g['root_N'][e] / g['root_N'][e].sum())


@pytest.mark.parametrize('name', ['mcts_deep_s10', 'mcts_prior_s10', 'mcts_deep_s10_thr', 'mcts_deep_s10_thr04'])
def test_planners_at_benchmark_depth_vs_reference(golden, models, name):
    """BASELINE configs[2] pinned at its real depth (mcts_deep_s10: 50 iterations x 10-sample expansions x depth-5 simulations, every
    iteration run: 205-node trees, paths up to 6 actions, path trimming on long paths) and the prior-exploration bonus combined with
    10-sample expansions and use_habit (mcts_prior_s10), both captured from the reference planner (oracle/make_golden_deep.py):
    the single-episode API on each episode alone AND the lock-step planner on all episodes together."""
    import daimc_amd
    g = golden(name)
    m = inject(_model(g, models))
    p = _deep_params(g)
    E = int(g['episodes'])
    frames = torch.from_numpy(g['frames'])
    m._stage = int(g['stage'])
    out, visits = daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
    for e in range(E):
        _check_deep(g, e, out[e], visits[e])
    # episode 0 through the reference-shaped single-episode API (Node / active_inference_mcts: all its noise rows start at 0) ...
    m._stage = int(g['stage'])
    p.host_tree = True
    _check_deep(g, 0, daimc_amd.active_inference_mcts(m, frames[0], p, o_shape=(1, 64, 64)))
    p.host_tree = False
    # ... and every episode planned alone by the lock-step planner at its global episode offset
    for e in range(E):
        m._stage = int(g['stage'])
        out1, v1 = daimc_amd.active_inference_mcts_batch(m, frames[e:e + 1], p, o_shape=(1, 64, 64), episode_offset=e)
        _check_deep(g, e, out1[0], v1[0])


def _fixture_params(g):
    """MCTS_Params with every field the new fixtures store (oracle/make_golden_defaults.py)"""
    import daimc_amd
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.simulation_repeats = int(g['repeats']), int(g['simulation_depth']), int(g['simulation_repeats'])
    p.use_means, p.threshold, p.C, p.samples = bool(g['use_means']), float(g['threshold']), float(g['C']), int(g['samples'])
    p.using_prior_for_exploration, p.use_habit = bool(g['using_prior_for_exploration']), bool(g['use_habit'])
    return p


@pytest.mark.parametrize('name', ['mcts_defaults', 'mcts_defaults_full', 'mcts_simrep2_s10'])
def test_planner_at_the_reference_defaults_and_with_two_simulations(golden, models, name):
    """The reference planner's OWN default call (mcts.py:139-148: 300 repeats, simulation depth 3, use_means -> calculate_G_mean expansions,
    threshold 0.5; `MCTS_Params()` untouched on both sides) and simulation_repeats = 2 (mcts.py:185-189), captured from the reference
    (oracle/make_golden_defaults.py).  mcts_defaults: six episodes that stop before iterations 300 (never), 217, 177, 125, 65, 21;
    mcts_defaults_full: the stop out of reach, 1 205-node trees = the lock-step planner's node capacity, 300-entry path history;
    mcts_simrep2_s10: 10-sample expansions, depth-5 simulations, two simulations per iteration, one early stop.
    Fixture episode k is GLOBAL episode episode_ids[k] of the batch synth.make_frames(frame_seed, n_frames) (the episodes were picked from
    a 40-frame probe for decision margins >= 5e-5: a 300-iteration episode takes ~2 000 argmax decisions, and an fp32-level difference in
    G -- 1e-3 -- moves a score by ~1e-5).  Compared: repeats_done, states_explored, every path, the G history, the final path, the root
    visit counts -- through the lock-step planner on the WHOLE batch (the other episodes plan alongside) and each fixture episode alone
    on the device planner at its global episode offset."""
    import daimc_amd
    g = golden(name)
    m = inject(_model(g, models))
    p = _fixture_params(g)
    if name == 'mcts_defaults':
        d = daimc_amd.MCTS_Params()
        assert all(getattr(p, k) == getattr(d, k) for k in vars(d)), 'the fixture is the reference default call'
        p = d                                                     # literally the untouched default object
    ids = [int(i) for i in g['episode_ids']]
    batch = synth.make_frames(int(g['frame_seed']), int(g['n_frames']))
    assert np.array_equal(batch[ids], g['frames'])
    E = batch.shape[0]
    m._stage = int(g['stage'])
    out, visits = daimc_amd.active_inference_mcts_batch(m, torch.from_numpy(batch), p, o_shape=(1, 64, 64))
    planner = next(pl for pl in m._planners.values() if pl.E == E and pl.p.repeats == p.repeats and pl.p.simulation_repeats == p.simulation_repeats)
    assert planner.cap == 1 + 4 * (p.repeats + 2) and int(planner.n_nodes.max()) <= planner.cap
    for k, e in enumerate(ids):
        _check_deep(g, k, out[e], visits[e])
    assert int(planner.n_nodes[ids].max()) == int(g['n_nodes'].max())
    # the longest-running and the earliest-stopping fixture episode alone on the device planner (the reference's one-episode call shape)
    for k in sorted({int(np.argmax(g['repeats_done'])), int(np.argmin(g['repeats_done']))}):
        m._stage = int(g['stage'])
        out1, v1 = daimc_amd.active_inference_mcts_batch(m, torch.from_numpy(g['frames'][k:k + 1]), p, o_shape=(1, 64, 64), episode_offset=ids[k])
        _check_deep(g, k, out1[0], v1[0])
    if ids[0] == 0:
        # episode 0 through the reference-shaped single-episode API on the host Node tree (all its noise rows start at 0)
        m._stage = int(g['stage'])
        p.host_tree = True
        _check_deep(g, 0, daimc_amd.active_inference_mcts(m, torch.from_numpy(g['frames'][0]), p, o_shape=(1, 64, 64)))
        p.host_tree = False


@pytest.mark.parametrize('name', ['mcts_defaults', 'mcts_simrep2_s10'])
def test_lockstep_batch_of_64_at_the_reference_defaults_contains_the_reference_episodes(golden, models, name):
    """the same captures inside a 64-episode lock-step batch (second stream + replica context, lagged host check, compaction of the episodes
    the default threshold stops): the fixture's episodes are bit-for-bit the reference's whatever the other episodes do"""
    import daimc_amd
    g = golden(name)
    m = inject(_model(g, models))
    p = _fixture_params(g)
    ids = [int(i) for i in g['episode_ids']]
    batch = synth.make_frames(int(g['frame_seed']), int(g['n_frames']))
    E = 64
    frames = torch.cat([torch.from_numpy(batch), torch.from_numpy(synth.make_frames(79, E - batch.shape[0]))], 0)
    m._stage = int(g['stage'])
    out, visits = daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
    planner = next(pl for pl in m._planners.values() if pl.E == E and pl.p.repeats == p.repeats and pl.p.simulation_repeats == p.simulation_repeats)
    assert planner.overlap and planner._ids is not None and len(planner._ids[1]) < E          # stopped episodes were compacted out
    for k, e in enumerate(ids):
        _check_deep(g, k, out[e], visits[e])
    np.testing.assert_allclose(visits.sum(1).numpy(), 1.0, rtol=1e-6)
    assert all(o_[2] == len(o_[3]) * p.simulation_depth * p.simulation_repeats for o_ in out)


def test_batched_mcts_episode_invariance(models):
    """episode e planned inside a batch of 3 == planned alone with episode_offset = e (global noise keys):
    the property that lets episodes shard across GPUs with identical results"""
    import daimc_amd
    m = models(1234, 1.15, 21)
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.use_means, p.threshold = 5, 3, False, 0.9
    p.samples = 2
    frames = torch.from_numpy(synth.make_frames(55, 3)[:, 0][:, None])
    m._stage = 0
    out3, dist3 = daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
    for e in range(3):
        m._stage = 0
        out1, dist1 = daimc_amd.active_inference_mcts_batch(m, frames[e:e + 1], p, o_shape=(1, 64, 64), episode_offset=e)
        assert out1[0][0] == out3[e][0] and out1[0][3] == out3[e][3]
        assert out1[0][4] == out3[e][4]
        assert torch.equal(dist1[0], dist3[e])


# ------------------------------------------------------------------------------------------------------
# more oracle comparisons: tile-tail sizes, the 10-sample path, trajectory API, checkpoint round trip
# ------------------------------------------------------------------------------------------------------
def test_calculate_G_ten_samples_vs_oracle(models, weights_cache):
    seed, st, M, S = 31, 2, 4, 10
    w = weights_cache(1234, 1.15)
    m = models(1234, 1.15, seed)
    orc = EO.OracleModel(w, EO.PhiloxNoise(seed))
    s0 = np.tile(PX.uniform_fill(8, (1, 10), 901, -1.0, 1.0), (M, 1))
    pi0 = np.eye(4, dtype=np.float32)
    with torch.no_grad():
        oG, oT, ops1, ops1m, opo1 = orc.calculate_G(torch.from_numpy(s0), torch.from_numpy(pi0), S, st)
    G, terms, ps1, ps1m, po1 = m.calculate_G(s0, pi0, samples=S, stage=st, eps=eps_calcG(seed, M, S, st))
    np.testing.assert_allclose(c(terms[0]), oT[0].numpy(), atol=1e-4)
    np.testing.assert_allclose(c(terms[1]), oT[1].numpy(), atol=1e-4)
    np.testing.assert_allclose(c(G), oG.numpy(), atol=gtol(orc.last_term2_parts[0].numpy()))
    np.testing.assert_allclose(c(po1), opo1.numpy(), rtol=1e-5, atol=1e-5)


def test_calculate_G_many_rows_vs_oracle(models, weights_cache):
    """M = 150, S = 4: 1800 decoder images (k_dec_a ticket queue), 600 encoder images (more than the persistent encoder
    workgroups) and 1200 transition rows in one call, against the oracle."""
    seed, st, M, S = 37, 6, 150, 4
    w = weights_cache(1234, 1.15)
    m = models(1234, 1.15, seed)
    orc = EO.OracleModel(w, EO.PhiloxNoise(seed))
    s0 = PX.uniform_fill(8, (M, 10), 904, -1.0, 1.0)
    pi0 = np.eye(4, dtype=np.float32)[np.arange(M) % 4]
    with torch.no_grad():
        oG, oT, ops1, ops1m, opo1 = orc.calculate_G(torch.from_numpy(s0), torch.from_numpy(pi0), S, st)
    G, terms, ps1, ps1m, po1 = m.calculate_G(s0, pi0, samples=S, stage=st, eps=eps_calcG(seed, M, S, st))
    np.testing.assert_allclose(c(terms[0]), oT[0].numpy(), atol=1e-4)
    np.testing.assert_allclose(c(terms[1]), oT[1].numpy(), atol=1e-4)
    np.testing.assert_allclose(c(G), oG.numpy(), atol=gtol(orc.last_term2_parts[0].numpy()))
    np.testing.assert_allclose(c(po1), opo1.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c(ps1), ops1.numpy(), rtol=1e-5, atol=2e-6)


def test_decoder_encoder_tile_tails_vs_oracle(models, weights_cache):
    """M = 257 rows: exercises the 64-row tiles of the dense kernels and the persistent image loops with a ragged tail"""
    seed, st, M = 41, 9, 257
    w = weights_cache(1234, 1.35)
    m = models(1234, 1.35, seed)
    orc = EO.OracleModel(w, EO.PhiloxNoise(seed))
    s = PX.uniform_fill(8, (M, 10), 902, -1.5, 1.5)
    with torch.no_grad():
        opo = orc.decoder(torch.from_numpy(s), PX.PASS_D1, 3, st)
        omean, olv = orc.encoder(opo, PX.PASS_E1, 3, st)
    po = m.model_down.decoder(s, stage=st, pass_=PX.PASS_D1, sample=3)
    np.testing.assert_allclose(c(po), opo.numpy(), rtol=1e-5, atol=1e-5)
    _, mean, lv = m.model_down.encoder_with_sample(opo.numpy(), stage=st, pass_=PX.PASS_E1, sample=3)
    # gain 1.35 drives hidden activations to O(10): fp32 summation-order noise is ~1e-6 * that
    np.testing.assert_allclose(c(mean), omean.numpy(), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(c(lv), olv.numpy(), rtol=1e-5, atol=2e-5)


def test_decoder_many_rows_vs_oracle(models, weights_cache):
    """M = 1100 rows: more images than persistent workgroup slots, so k_dec_a claims images from its ticket counter and the
    persistent k_fc4 ranges cross row tiles -- the paths the throughput bench runs, checked against the oracle."""
    seed, st, M = 43, 4, 1100
    w = weights_cache(1234, 1.15)
    m = models(1234, 1.15, seed)
    orc = EO.OracleModel(w, EO.PhiloxNoise(seed))
    s = PX.uniform_fill(9, (M, 10), 903, -1.5, 1.5)
    with torch.no_grad():
        opo = orc.decoder(torch.from_numpy(s), PX.PASS_D2A, 1, st)
    po = m.model_down.decoder(s, stage=st, pass_=PX.PASS_D2A, sample=1)
    np.testing.assert_allclose(c(po), opo.numpy(), rtol=1e-5, atol=1e-5)
    po2 = m.model_down.decoder(s, stage=st, pass_=PX.PASS_D2A, sample=1)
    assert torch.equal(po, po2)          # the dynamic image schedule does not change any bit


@pytest.mark.parametrize('M', [37, 16400])
def test_fused_dense_heads_equal_layerwise_launches(models, weights_cache, M):
    """k_head (the decoder's three 256-wide layers / the encoder's four dense layers as one launch, activations in LDS; 16 rows per
    workgroup, 32 from 16384 rows up) against the same layers as k_dense launches (option head_unfused = 1): the same MC-dropout
    masks, values equal up to the fp32 summation order of the two contractions.  Ragged tails in both tile sizes."""
    seed, st = 47, 6
    m = models(1234, 1.15, seed)
    s = PX.uniform_fill(12, (M, 10), 905, -1.5, 1.5)
    po1 = m.model_down.decoder(s, stage=st, pass_=PX.PASS_D2B, sample=2)
    _, mean1, lv1 = m.model_down.encoder_with_sample(po1, stage=st, pass_=PX.PASS_E1, sample=2)
    m.set_option('head_unfused', 1)
    try:
        po0 = m.model_down.decoder(s, stage=st, pass_=PX.PASS_D2B, sample=2)
        _, mean0, lv0 = m.model_down.encoder_with_sample(po1, stage=st, pass_=PX.PASS_E1, sample=2)
    finally:
        m.set_option('head_unfused', 0)
    np.testing.assert_allclose(c(po1), c(po0), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c(mean1), c(mean0), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(c(lv1), c(lv0), rtol=1e-5, atol=2e-5)


def test_given_trajectory_vs_oracle(models, weights_cache):
    seed, st, T = 51, 4, 5
    w = weights_cache(1234, 1.15)
    m = models(1234, 1.15, seed)
    orc = EO.OracleModel(w, EO.PhiloxNoise(seed))
    s0 = PX.uniform_fill(8, (T, 10), 903, -1, 1); ps1 = PX.uniform_fill(8, (T, 10), 904, -1, 1)
    mean = PX.uniform_fill(8, (T, 10), 905, -1, 1); lv = PX.uniform_fill(8, (T, 10), 906, -2, 0)
    pi0 = np.eye(4, dtype=np.float32)[[0, 3, 1, 2, 2]]
    with torch.no_grad():
        oG = orc.calculate_G_given_trajectory(*(torch.from_numpy(x) for x in (s0, ps1, mean, lv, pi0)), st)
    eps = np.stack([np.zeros((T, 10), np.float32), PX.normals(seed, T, 10, PX.PASS_T2, 0, st), PX.normals(seed, T, 10, PX.PASS_D2B, 0, st)])
    G = m.calculate_G_given_trajectory(s0, ps1, mean, lv, pi0, stage=st, eps=eps)
    np.testing.assert_allclose(c(G), oG.numpy(), atol=gtol(np.array([2800.0])))


def test_check_reward_and_helpers(models, golden):
    g = golden('helpers')
    m = models(1234, 1.15, 1)
    np.testing.assert_allclose(c(m.check_reward(g['p'])), g['reward'], rtol=2e-6)
    q = m.habitual_net(g['p'])
    assert q.shape == (3, 4) and torch.allclose(q.sum(1), torch.ones(3, device=q.device), atol=1e-6)
    fut = m.imagine_future_from_o(g['p'], np.eye(4, dtype=np.float32)[:3])
    assert fut.shape == (3, 1, 64, 64) and torch.isfinite(fut).all()


def test_checkpoint_round_trip(models, tmp_path):
    """save_weights / load_weights with the reference's file names and state_dict keys (torchmodel.py:167-177)"""
    import daimc_amd
    m = models(1234, 1.15, 3)
    m.save_weights(str(tmp_path))
    for f in ('checkpoint_down.pth', 'checkpoint_top.pth', 'checkpoint_mid.pth'):
        assert (tmp_path / f).exists()
    sd = torch.load(tmp_path / 'checkpoint_down.pth')
    assert sd['po_net.13.weight'].shape == (64, 64, 3, 3) and sd['qs_net.9.weight'].shape == (256, 576)
    m2 = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=3, init_weights=False)
    m2.load_weights(str(tmp_path))
    s = PX.uniform_fill(8, (4, 10), 907, -1, 1)
    a = m.model_down.decoder(s, stage=0)
    b = m2.model_down.decoder(s, stage=0)
    assert torch.equal(a, b)
    bad = dict(sd); bad['qs_net.9.weight'] = torch.zeros(256, 256)
    with pytest.raises(ValueError, match='torchmodel.py:94'):
        m2.model_down.load_state_dict(bad)


def test_api_errors_are_loud(models):
    m = models(1234, 1.15, 1)
    with pytest.raises(RuntimeError):
        m.calculate_G(np.zeros((4, 10), np.float32), np.eye(4, dtype=np.float32), samples=0)
    with pytest.raises(ValueError):
        m.calculate_G_4_repeated(np.zeros((3, 1, 64, 64), np.float32))
    with pytest.raises(ValueError):
        import daimc_amd
        daimc_amd.ActiveInferenceModel(10, 9, 0.0, 1.0, 1.0)          # pi_dim outside 2..6
    import daimc_amd
    p = daimc_amd.MCTS_Params()
    assert daimc_amd.active_inference_mcts(m, [], p) == ([0], 0, 0, [], [])


def test_plan_actions_batch_vs_golden(golden, models):
    """model side of make_batch_dsprites_active_inference (util.py:55-70): posterior of the summed EFE vs the reference's"""
    import daimc_amd
    g = golden('rollout_m8d2s2')
    m = _model(g, models)
    frames = g['o'][::4]                                  # the golden rows are (frame i, action a) at 4i + a
    eps = eps_rollout(int(g['nseed']), 8, 2, 2, int(g['stage']))
    pi0, logP, P, sumG = daimc_amd.plan_actions_batch(m, frames, deepness=2, samples=2, stage=int(g['stage']), eps=eps)
    np.testing.assert_allclose(c(sumG).reshape(-1), g['sum_G'], atol=2 * gtol(np.array([2800.0])))
    np.testing.assert_allclose(c(P), g['Ppi'], atol=5e-3)
    assert pi0.shape == (2, 4) and torch.all(pi0.sum(1) == 1)


def test_full_depth_rollout_vs_oracle(models, weights_cache):
    """BASELINE cfg-2 depth and sample count (D=5, S=10) on 4 rows against the CPU oracle"""
    seed, st, M, D, S = 61, 7, 4, 5, 10
    w = weights_cache(1234, 1.15)
    m = models(1234, 1.15, seed)
    orc = EO.OracleModel(w, EO.PhiloxNoise(seed))
    o = np.repeat(synth.make_frames(71, 1), 4, axis=0)
    pi = np.eye(4, dtype=np.float32)
    with torch.no_grad():
        oG, oT, opo1 = orc.calculate_G_repeated(torch.from_numpy(o), torch.from_numpy(pi), D, False, S, st)
    G, T, po1 = m.calculate_G_repeated(o, pi, steps=D, samples=S, stage=st, eps=eps_rollout(seed, M, D, S, st))
    np.testing.assert_allclose(c(T[0]), oT[0].numpy(), atol=5e-4)
    np.testing.assert_allclose(c(T[1]), oT[1].numpy(), atol=5e-4)
    np.testing.assert_allclose(c(G), oG.numpy(), atol=D * gtol(np.array([2800.0])))
    np.testing.assert_allclose(c(po1), opo1.numpy(), rtol=1e-5, atol=5e-5)


def test_simulate_batch_vs_oracle_per_episode(models, weights_cache):
    """E lock-step episodes == E independent reference-style simulations (global episode keys)"""
    seed, st, E, T = 71, 12, 3, 4
    w = weights_cache(1234, 1.15)
    m = inject(models(1234, 1.15, seed))
    starts = PX.uniform_fill(8, (E, 10), 910, -1, 1)
    G, pi0, q0 = m.simulate_batch(starts, T, use_means=False, stage=st, row_offset=5)
    orc = EO.OracleModel(w, EO.PhiloxNoise(seed))
    for e in range(E):
        with torch.no_grad():
            oG, opi0, oq = orc.mcts_step_simulate(torch.from_numpy(starts[e]), T, False, st, episode=5 + e)
        assert np.array_equal(c(pi0[e]), opi0.numpy())
        np.testing.assert_allclose(c(q0[e]), oq.numpy(), rtol=1e-5, atol=1e-6)
        assert abs(float(G[e]) - oG) < gtol(np.array([2800.0]))


def test_mcts_habit_shortcut_and_prior_exploration(models):
    import daimc_amd
    m = models(1234, 1.15, 81)
    frame = torch.from_numpy(synth.make_frames(91, 1)[0])
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.threshold = 4, 2, 0.9
    p.use_habit = True
    p.threshold = -1.0                     # habit posterior always "confident": decision in phase A (mcts.py:166-170)
    m._stage = 0
    path, reps, explored, ap, ag = daimc_amd.active_inference_mcts(m, frame, p, o_shape=(1, 64, 64))
    assert len(path) == 1 and 0 <= path[0] < 4 and reps == 0 and explored == 0
    p.use_habit, p.threshold, p.using_prior_for_exploration = False, 0.9, True
    m._stage = 0
    path, reps, explored, ap, ag = daimc_amd.active_inference_mcts(m, frame, p, o_shape=(1, 64, 64))
    assert reps == 4 and explored == 8 and len(ap) == 4 and all(np.isfinite(ag))
    m._stage = 0
    out, dist = daimc_amd.active_inference_mcts_batch(m, frame[None], p, o_shape=(1, 64, 64))
    assert out[0][0] == path and out[0][3] == [[int(a) for a in q] for q in ap]


def test_c_abi_from_plain_c(tmp_path):
    """compile and run tests/c_abi_smoke.c: the boundary is usable from C with nothing but the header and the .so"""
    import subprocess
    from conftest import ROOT
    import os
    pkg = os.path.join(ROOT, 'deep-active-inference-mc_amd')
    exe = str(tmp_path / 'c_abi_smoke')
    cmd = ['gcc', os.path.join(ROOT, 'tests', 'c_abi_smoke.c'), '-I' + os.path.join(ROOT, 'include'), '-I/opt/rocm/include',
           '-D__HIP_PLATFORM_AMD__', '-L' + pkg, '-lefe_mi355x', '-L/opt/rocm/lib', '-lamdhip64', '-lm',
           '-Wl,-rpath,' + pkg, '-Wl,-rpath,/opt/rocm/lib', '-o', exe]
    subprocess.run(cmd, check=True)
    # the weights the committed fixture blob was captured with, as a flat file the C program loads through efe_set_weight
    g = load_golden('calcG_m4s1_g115')
    wfile = str(tmp_path / 'weights.bin')
    with open(wfile, 'wb') as f:
        for key, arr in synth.make_weights(int(g['wseed']), float(g['gain'])).items():
            arr = np.ascontiguousarray(arr, dtype='<f4')
            f.write(key.encode() + b'\0' + np.int32(arr.ndim).tobytes() + np.asarray(arr.shape, dtype='<i8').tobytes() + arr.tobytes())
    blob = os.path.join(ROOT, 'tests', 'golden', 'calcG_m4s1_g115.bin')
    r = subprocess.run([exe, wfile, blob], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'c_abi_smoke OK' in r.stdout and 'c_abi_smoke fixture OK' in r.stdout


def test_reparameterize_through_engine(models):
    m = models(1234, 1.15, 17)
    mean = PX.uniform_fill(8, (6, 10), 920, -1, 1); lv = PX.uniform_fill(8, (6, 10), 921, -2, 0)
    eps = PX.normals(17, 6, 10, PX.PASS_ROOT, 0, 3)
    ref = eps * np.exp(lv * np.float32(0.5)) + mean
    np.testing.assert_allclose(c(m.model_down.reparameterize(mean, lv, stage=3, eps=eps)), ref, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(c(m.model_mid.reparameterize(mean, lv, stage=3)), ref, rtol=1e-4, atol=1e-5)   # device Box-Muller


def test_mcts_tree_kernels_vs_host_torch(models):
    """efe_mcts_select / efe_mcts_stop against the same formulas evaluated with torch on the host (the reference's
    operation order, mcts.py:36-57, 130-131), on random trees that include unvisited edges (N = 0 -> inf / NaN scores)."""
    import ctypes as C
    from daimc_amd import _lib
    m = models(1234, 1.15, 21)
    e = m._ready()
    E, A, cap, sd, depth = 37, 4, 21, 10, 6
    g = torch.Generator().manual_seed(5)
    W = torch.randn(E, cap, A, generator=g) * 3
    N = torch.randint(0, 4, (E, cap, A), generator=g).float()
    N[:, 0] = torch.randint(1, 6, (E, A), generator=g).float()          # roots are always visited
    N[3, 0] = 0.0; W[3, 0] = 0.0                                        # one all-NaN root (0/0 everywhere)
    Qpi = torch.rand(E, cap, A, generator=g)
    child = torch.full((E, cap, A), -1, dtype=torch.int32)
    for ep in range(E):                                                 # a random tree: node k expanded with prob. 0.6 while slots last
        nxt = 1
        for node in range(cap):
            if node >= nxt:
                break
            if (node == 0 or torch.rand(1, generator=g).item() < 0.6) and nxt + A <= cap:
                child[ep, node] = torch.arange(nxt, nxt + A, dtype=torch.int32)
                nxt += A
    S = torch.randn(E, cap, sd, generator=g)
    active = (torch.rand(E, generator=g) < 0.8).to(torch.uint8)
    dev = m.device
    dW, dN, dQ, dC, dS, dA = (t.to(dev).contiguous() for t in (W, N, Qpi, child, S, active))
    tree = _lib.EfeMctsTree(dW.data_ptr(), dN.data_ptr(), dQ.data_ptr(), dC.data_ptr(), dS.data_ptr(), E, cap, A, sd)
    p = lambda t: C.c_void_p(t.data_ptr())
    for use_prior in (0, 1):
        pn = torch.zeros(E, depth, dtype=torch.int32, device=dev); pa = torch.zeros_like(pn)
        pl = torch.zeros(E, dtype=torch.int32, device=dev); leaf = torch.zeros_like(pl)
        ls = torch.zeros(E, sd, device=dev); lr = torch.zeros(E * A, sd, device=dev)
        e.check(e.lib.efe_mcts_select(e.ctx, C.byref(tree), p(dA), 1.5, use_prior, depth, p(pn), p(pa), p(pl), p(leaf), p(ls), p(lr), e.stream()))
        for ep in range(E):
            cur, path = 0, []
            if active[ep]:
                while True:
                    q = W[ep, cur] / N[ep, cur]
                    q = q - q.min()
                    q = q / q.sum()
                    bonus = 1.5 * 1.0 / N[ep, cur]
                    if use_prior:
                        bonus = 1.5 * Qpi[ep, cur] * 1.0 / N[ep, cur]      # mcts.py:45, left to right
                    a = int(torch.argmax(q + bonus))
                    path.append((cur, a))
                    cur = int(child[ep, cur, a])
                    if child[ep, cur, 0] < 0:
                        break
            assert int(pl[ep]) == len(path) and int(leaf[ep]) == cur, (ep, use_prior)
            assert [(int(pn[ep, d]), int(pa[ep, d])) for d in range(len(path))] == path, (ep, use_prior)
            assert torch.equal(ls[ep].cpu(), S[ep, cur]) and torch.equal(lr[ep * A + A - 1].cpu(), S[ep, cur])
    # early-stop test
    stop_at = torch.full((E,), -1, dtype=torch.int32, device=dev); n_act = torch.zeros(1, dtype=torch.int32, device=dev)
    act2 = dA.clone()
    e.check(e.lib.efe_mcts_stop(e.ctx, C.byref(tree), p(act2), p(stop_at), 7, 0.2, p(n_act), e.stream()))
    dist = N[:, 0] / N[:, 0].sum(dim=1, keepdim=True)
    done = active.bool() & ((dist.max(dim=1).values - dist.mean(dim=1)) > 0.2)
    assert torch.equal(act2.cpu().bool(), active.bool() & ~done)
    assert torch.equal(stop_at.cpu() == 7, done) and int(n_act.item()) == int((active.bool() & ~done).sum())


def test_mcts_step_equals_backprop_stop_select(models):
    """efe_mcts_step (ONE launch between two iterations' engine calls) == efe_mcts_expand of the previous iteration's leaf, efe_mcts_backprop of
    the previous iteration, then efe_mcts_stop, then efe_mcts_select, on random trees: same tree statistics, children, node states and counts,
    history row, active set, stop iterations, active count and selection"""
    import ctypes as C
    from daimc_amd import _lib
    m = models(1234, 1.15, 21)
    e = m._ready()
    E, A, cap, sd, depth, R = 41, 4, 25, 10, 6, 2
    g = torch.Generator().manual_seed(11)
    dev = m.device
    p = lambda t: C.c_void_p(t.data_ptr())

    def fresh():
        gg = torch.Generator().manual_seed(12)
        W = torch.randn(E, cap, A, generator=gg) * 3
        N = torch.randint(0, 4, (E, cap, A), generator=gg).float()
        N[:, 0] = torch.randint(1, 6, (E, A), generator=gg).float()
        Qpi = torch.rand(E, cap, A, generator=gg)
        child = torch.full((E, cap, A), -1, dtype=torch.int32)
        nn_ = torch.zeros(E, dtype=torch.int32)
        for ep in range(E):
            nxt = 1
            for node in range(cap):
                if node >= nxt:
                    break
                if (node == 0 or torch.rand(1, generator=gg).item() < 0.6) and nxt + A <= cap - A:      # (room for one more expansion)
                    child[ep, node] = torch.arange(nxt, nxt + A, dtype=torch.int32)
                    nxt += A
            nn_[ep] = nxt
        S = torch.randn(E, cap, sd, generator=gg)
        active = (torch.rand(E, generator=gg) < 0.85).to(torch.uint8)
        t = [x.to(dev).contiguous() for x in (W, N, Qpi, child, S, active, nn_)]
        tree = _lib.EfeMctsTree(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(), E, cap, A, sd)
        bufs = dict(pn=torch.zeros(E, depth, dtype=torch.int32, device=dev), pa=torch.zeros(2, E, depth, dtype=torch.int32, device=dev),
                    pl=torch.zeros(2, E, dtype=torch.int32, device=dev), leaf=torch.zeros(E, dtype=torch.int32, device=dev),
                    ls=torch.zeros(E, sd, device=dev), lr=torch.zeros(E * A, sd, device=dev), stop=torch.full((E,), -1, dtype=torch.int32, device=dev),
                    nact=torch.zeros(2, dtype=torch.int32, device=dev), g=torch.zeros(E, device=dev), hact=torch.zeros(E, dtype=torch.uint8, device=dev))
        return t, tree, bufs
    sims = torch.randn(R, E, generator=g).to(dev)
    q0 = torch.rand(E, A, generator=g).to(dev)
    Gx = torch.randn(E * A, generator=g).to(dev)                    # the previous iteration's expansion results
    psx = torch.randn(E * A, sd, generator=g).to(dev)
    out = []
    for fused in (False, True):
        t, tree, b = fresh()
        act = t[5]
        # iteration 0: selection (no previous iteration)
        if fused:
            e.check(e.lib.efe_mcts_step(e.ctx, C.byref(tree), None, None, None, 1, None, None, None, p(act), p(b['stop']), 0, 0.25, p(b['nact'][0:]),
                                        1.5, 1, depth, p(b['pn']), p(b['pa'][0]), p(b['pl'][0]), p(b['leaf']), p(b['ls']), p(b['lr']), None, None, None, e.stream()))
            e.check(e.lib.efe_mcts_step(e.ctx, C.byref(tree), p(b['pa'][0]), p(b['pl'][0]), p(sims), R, p(q0), p(b['g']), p(b['hact']), p(act), p(b['stop']),
                                        1, 0.25, p(b['nact'][1:]), 1.5, 1, depth, p(b['pn']), p(b['pa'][1]), p(b['pl'][1]), p(b['leaf']), p(b['ls']), p(b['lr']),
                                        p(t[6]), p(Gx), p(psx), e.stream()))
        else:
            for it in (0, 1):
                if it == 1:
                    e.check(e.lib.efe_mcts_expand(e.ctx, C.byref(tree), p(t[6]), p(b['leaf']), p(act), p(Gx), p(psx), e.stream()))
                    e.check(e.lib.efe_mcts_backprop(e.ctx, C.byref(tree), p(b['pn']), p(b['pa'][0]), p(b['pl'][0]), p(b['leaf']), p(act), p(sims), R, p(q0),
                                                    depth, p(b['g']), p(b['hact']), e.stream()))
                na = torch.zeros(1, dtype=torch.int32, device=dev)
                e.check(e.lib.efe_mcts_stop(e.ctx, C.byref(tree), p(act), p(b['stop']), it, 0.25, p(na), e.stream()))
                b['nact'][it] = na[0]
                e.check(e.lib.efe_mcts_select(e.ctx, C.byref(tree), p(act), 1.5, 1, depth, p(b['pn']), p(b['pa'][it]), p(b['pl'][it]), p(b['leaf']),
                                              p(b['ls']), p(b['lr']), e.stream()))
        torch.cuda.synchronize()
        out.append([x.cpu() for x in t[:3]] + [act.cpu()] + [b[k].cpu() for k in ('pn', 'pa', 'pl', 'leaf', 'ls', 'lr', 'stop', 'nact', 'g', 'hact')] + [t[3].cpu(), t[4].cpu(), t[6].cpu()])
    for x, y in zip(*out):
        assert torch.equal(x, y) or (torch.isnan(x) == torch.isnan(y)).all() and torch.equal(torch.nan_to_num(x), torch.nan_to_num(y))
    assert int(out[0][11][1]) < int(out[0][11][0]) or int(out[0][11][0]) < E        # the stop test did something


# ------------------------------------------------------------------------------------------------------
# the benchmarked configuration itself, pinned at full size
# ------------------------------------------------------------------------------------------------------
def test_full_size_cfg2_vs_oracle(models, weights_cache):
    """BASELINE configs[1] exactly as bench.py runs it -- 128 rows (32 roots x 4 actions), depth 5, 10 MC samples, one
    efe_rollout call -- against the CPU oracle on the same seeded inputs (about 20 s of host time)."""
    seed, st, M, D, S = 1, 0, 128, 5, 10
    w = weights_cache(1234, 1.15)
    m = models(1234, 1.15, seed)
    orc = EO.OracleModel(w, EO.PhiloxNoise(seed))
    o = np.repeat(synth.make_frames(41, M // 4), 4, axis=0)
    pi = np.tile(np.eye(4, dtype=np.float32), (M // 4, 1))
    torch.set_num_threads(usable_cores())
    with torch.no_grad():
        oG, oT, opo1 = orc.calculate_G_repeated(torch.from_numpy(o), torch.from_numpy(pi), D, False, S, st)
    G, T, po1 = m.calculate_G_repeated(o, pi, steps=D, samples=S, stage=st, eps=eps_rollout(seed, M, D, S, st))
    np.testing.assert_allclose(c(T[0]), oT[0].numpy(), atol=5e-4)
    np.testing.assert_allclose(c(T[1]), oT[1].numpy(), atol=5e-4)
    np.testing.assert_allclose(c(G), oG.numpy(), atol=D * gtol(np.array([2800.0])))
    np.testing.assert_allclose(c(po1), opo1.numpy(), rtol=1e-5, atol=5e-5)
    P, _ = m.action_posterior(G)
    oP, _ = EO.softmax_multi_with_log(-oG.numpy(), 4)
    np.testing.assert_allclose(c(P), oP, atol=2e-3)
    # the same call in PRODUCTION noise mode (normals generated on the device from the same Philox stream, nothing injected): the only
    # difference from the mirror is Box-Muller's libm (an ulp on each normal), a few 1e-3 on G per stage -- at the benchmark's size
    Gd, Td, po1d = m.calculate_G_repeated(o, pi, steps=D, samples=S, stage=st)
    np.testing.assert_allclose(c(Td[0]), oT[0].numpy(), atol=2e-3)
    np.testing.assert_allclose(c(Td[1]), oT[1].numpy(), atol=2e-3)
    np.testing.assert_allclose(c(Gd), oG.numpy(), atol=D * (gtol(np.array([2800.0])) + 4e-3))
    np.testing.assert_allclose(c(po1d), opo1.numpy(), rtol=1e-4, atol=2e-4)


def test_custom_op_boundary_refuses_stale_handles_and_foreign_devices(weights_cache):
    """torch.ops.efe.* take the engine context as an integer handle: a handle that is not a live context (destroyed, made up) is a
    RuntimeError from the registry check (efe_ctx_alive), not a dereference of freed memory; a tensor on another device than the context's
    is refused; and a call leaves the caller's current HIP device as it found it (include/efe_engine.h, ABI 6)."""
    import ctypes as C
    import gc
    import daimc_amd
    from daimc_amd import _lib
    lib, ops = _lib.load(), _lib.load_ops()
    m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=3, init_weights=False)
    m.load_flat_weights(weights_cache(1234, 1.15))
    s = torch.zeros(4, 10, device='cuda:0')
    h = m._ready().h
    assert lib.efe_ctx_alive(C.c_void_p(h)) == 1
    q_live = ops.habit(h, s)[1]
    assert torch.isfinite(q_live).all()
    del m
    gc.collect()
    assert lib.efe_ctx_alive(C.c_void_p(h)) == 0                   # the context left the registry when it was destroyed
    for bad in (h, h + 64, 0xdead0000, -1):
        with pytest.raises(RuntimeError):
            ops.habit(bad, s)
    with pytest.raises(RuntimeError):
        ops.habit(0, s)
    # the C ABI itself refuses them too (return code 1, "stale or invalid context handle"), and a second destroy is a no-op
    assert lib.efe_habit(C.c_void_p(h), C.c_void_p(s.data_ptr()), 4, None, None, None, None) == 1
    assert b'stale' in lib.efe_last_error(C.c_void_p(h))
    lib.efe_destroy(C.c_void_p(h))
    # device discipline
    m2 = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=3, init_weights=False)
    m2.load_flat_weights(weights_cache(1234, 1.15))
    with pytest.raises(NotImplementedError):
        ops.habit(m2._ready().h, torch.zeros(4, 10))                # CPU tensors: no CPU kernel is registered (no fallback)
    if torch.cuda.device_count() >= 2:
        with pytest.raises(RuntimeError, match='device'):
            ops.habit(m2._ready().h, torch.zeros(4, 10, device='cuda:1'))
        m3 = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:1', seed=3, init_weights=False)
        m3.load_flat_weights(weights_cache(1234, 1.15))
        torch.cuda.set_device(0)
        q3 = m3.model_top.encode_s(torch.zeros(4, 10, device='cuda:1'))[1]
        assert torch.cuda.current_device() == 0                    # the call ran on GPU 1 and put the caller's device back
        assert torch.equal(q3.cpu(), q_live.cpu())
    dev = C.c_int(-1)
    assert lib.efe_get_device(m2._engine.ctx, C.byref(dev), None, 0) == 0 and dev.value == 0


def test_reserve_no_growth(models):
    """efe_reserve + efe_rollout_scratch_bytes: after reserving for a rollout size, calls at that size never grow the arena"""
    import daimc_amd
    m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=2)
    need = m.reserve(24, 3, 4)
    st0 = m.arena_stats()
    assert st0['capacity_bytes'] >= need
    o = synth.make_frames(35, 24)
    pi = np.eye(4, dtype=np.float32)[np.arange(24) % 4]
    for k in range(3):
        m.calculate_G_repeated(o, pi, steps=3, samples=4, stage=10 * k)
    torch.cuda.synchronize()
    st1 = m.arena_stats()
    assert st1['grow_count'] == st0['grow_count'] and st1['capacity_bytes'] == st0['capacity_bytes']
    assert 0 < st1['high_water_bytes'] <= need


def test_stream_switch_is_ordered(models):
    """one context, calls alternating between two torch streams: the engine orders a call behind the previous call's
    stream (the scratch arena is shared), so results equal the single-stream ones"""
    m = models(1234, 1.15, 14)
    o = synth.make_frames(36, 12)
    pi = np.eye(4, dtype=np.float32)[np.arange(12) % 4]
    ref = [m.calculate_G_repeated(o, pi, steps=2, samples=3, stage=4 * k)[0].clone() for k in range(4)]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ot, pit = torch.from_numpy(o).cuda(), torch.from_numpy(pi).cuda()
    torch.cuda.synchronize()
    got = []
    for k in range(4):
        with torch.cuda.stream(s1 if k % 2 == 0 else s2):
            got.append(m.calculate_G_repeated(ot, pit, steps=2, samples=3, stage=4 * k)[0])
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


def test_recommit_updates_one_tensor_without_leaking(models):
    import ctypes as C
    import daimc_amd
    m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=4)
    s = PX.uniform_fill(8, (4, 10), 930, -1, 1)
    a = m.model_down.decoder(s, stage=0)
    e = m._ready()
    # C-ABI user: change ONE tensor and commit again (the host copies of the others are kept)
    b4 = np.array([0.25], dtype=np.float32)
    shape = (C.c_int64 * 1)(1)
    e.check(e.lib.efe_set_weight(e.ctx, b'down.po_net.19.bias', b4.ctypes.data_as(C.c_void_p), shape, 1))
    e.check(e.lib.efe_commit_weights(e.ctx))
    b = m.model_down.decoder(s, stage=0)
    assert not torch.equal(a, b)
    sd = m.model_down.state_dict(); sd['po_net.19.bias'] = torch.tensor([0.25])
    m.model_down.load_state_dict(sd)
    assert torch.equal(m.model_down.decoder(s, stage=0), b)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(6):                                   # 6 x 21 MB of packed weights would show as 128 MB
        m.model_down.load_state_dict(sd)
        m.model_down.decoder(s, stage=0)
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < 32 * 2 ** 20


def test_torch_ops_are_the_dispatch_path(models):
    """torch.ops.efe.* (csrc/torch_ops.cpp): registered schemas, called directly with device tensors, equal the mirror's results"""
    m = models(1234, 1.15, 19)
    e = m._ready()
    ops = torch.ops.efe
    for name in ('transition', 'decoder', 'encoder', 'habit', 'calculate_g', 'rollout', 'trajectory', 'simulate', 'action_posterior',
                 'check_reward', 'reparameterize'):
        assert hasattr(ops, name)
    s0 = torch.from_numpy(PX.uniform_fill(8, (4, 10), 940, -1, 1)).cuda()
    G, terms, ps1, ps1m, po1, parts = ops.calculate_g(e.h, s0, m.pi_one_hot, 3, False, 19, 7, 0, None)
    G2, t2, ps12, _, po12 = m.calculate_G(s0, m.pi_one_hot, samples=3, stage=7)
    assert torch.equal(G, G2) and torch.equal(po1, po12) and torch.equal(ps1, ps12)
    with pytest.raises(NotImplementedError):
        ops.habit(e.h, s0.cpu())                          # no CPU kernel is registered: no fallback
    with pytest.raises(RuntimeError):
        ops.calculate_g(e.h, s0, m.pi_one_hot, 0, False, 19, 7, 0, None)


# ------------------------------------------------------------------------------------------------------
# efe_rows.mask: early-stopped episodes are skipped by the per-image kernels, live rows do not change
# ------------------------------------------------------------------------------------------------------
def test_row_mask_live_rows_are_bit_identical(models):
    m = models(1234, 1.15, 33)
    A, Eps = 4, 5
    M = A * Eps
    s0 = PX.uniform_fill(2, (M, 10), 77, -1, 1)
    pi0 = np.eye(4, dtype=np.float32)[np.arange(M) % 4]
    starts = PX.uniform_fill(9, (Eps, 10), 78, -1, 1)
    ref = m.calculate_G(s0, pi0, samples=3, stage=4)
    rsim = m.simulate_batch(starts, 3, use_means=False, stage=9)
    alive = torch.tensor([1, 0, 1, 1, 0], dtype=torch.uint8, device=m.device)
    from daimc_amd.model import Rows
    out = m.calculate_G(s0, pi0, samples=3, stage=4, rows=Rows(mask=alive, rows_per_entry=A))
    osim = m.simulate_batch(starts, 3, use_means=False, stage=9, rows=Rows(mask=alive))
    rows = alive.bool().repeat_interleave(A)
    assert torch.equal(out[0][rows], ref[0][rows])                       # G
    for k in range(3):
        assert torch.equal(out[1][k][rows], ref[1][k][rows])             # terms
    assert torch.equal(out[2][rows], ref[2][rows]) and torch.equal(out[4][rows], ref[4][rows])      # ps1, po1
    ep = alive.bool()
    assert torch.equal(osim[0][ep], rsim[0][ep]) and torch.equal(osim[1][ep], rsim[1][ep])
    # the mask belonged to those calls only (ABI 6: no context state): a plain call evaluates everything again
    again = m.calculate_G(s0, pi0, samples=3, stage=4)
    assert torch.equal(again[0], ref[0])
    assert not hasattr(m, 'set_row_mask')
    with pytest.raises(ValueError):
        Rows(mask=torch.ones(5, dtype=torch.float32, device=m.device), rows_per_entry=A)


def test_batched_mcts_skipping_stopped_episodes_changes_nothing(models):
    """the lock-step planner with the engine skipping early-stopped episodes == the same planner evaluating every episode"""
    import daimc_amd
    m = models(1234, 1.15, 21)
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.use_means = 12, 3, False
    p.samples = 2
    frames = torch.from_numpy(synth.make_frames(56, 9)[:, 0][:, None])
    mixed = False
    for thr in (0.3, 0.25, 0.2, 0.15, 0.1):                 # the first threshold at which some, not all, episodes stop early
        p.threshold = thr
        res = []
        for skip in (True, False):
            p.skip_stopped = skip
            m._stage = 0
            res.append(daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64)))
        (out_a, dist_a), (out_b, dist_b) = res
        for e in range(len(out_a)):
            assert out_a[e][0] == out_b[e][0] and out_a[e][1] == out_b[e][1] and out_a[e][3] == out_b[e][3] and out_a[e][4] == out_b[e][4]
        assert torch.equal(dist_a, dist_b)
        stops = [o[1] for o in out_a]
        if min(stops) < max(stops):
            mixed = True
            break
    assert mixed, 'no threshold stopped some episodes early: the fixture does not exercise the skip'


def test_small_launch_image_split_is_bit_identical(models):
    """k_dec_b4 with four workgroups per image (decoder launches of <= 128 images: the one-episode planner) == one workgroup per image,
    bit for bit: the per-image sum is defined quarter-wise in both forms.  Checked on calculate_G (24 and 120 images), on a rollout and
    on simulate_batch, with the images that are stored, and against a launch above the threshold that contains the same rows."""
    m = models(1234, 1.15, 41)
    s4 = torch.from_numpy(PX.uniform_fill(5, (4, 10), 91, -1, 1)).to(m.device)
    starts = torch.from_numpy(PX.uniform_fill(6, (3, 10), 92, -1, 1)).to(m.device)
    o = np.repeat(synth.make_frames(42, 1), 4, axis=0)
    res = {}
    try:
        for split in (1, 0):
            m.set_option('dec_split', split)
            res[split] = (m.calculate_G(s4, m.pi_one_hot, samples=2, stage=3), m.calculate_G(s4, m.pi_one_hot, samples=10, stage=4),
                          m.calculate_G_repeated(o, np.eye(4, dtype=np.float32), steps=2, samples=3, stage=6),
                          m.simulate_batch(starts, 5, use_means=False, stage=9))
    finally:
        m.set_option('dec_split', 1)
    for a_, b_ in zip(res[1], res[0]):
        for x, y in zip(a_, b_):
            if isinstance(x, (list, tuple)):
                assert all(torch.equal(u, v) for u, v in zip(x, y))
            else:
                assert torch.equal(x, y)
    # the same four rows inside a 40-row call (400 images: one workgroup per image) -- rows are keyed globally, so rows 0..3 are the same draws
    big = torch.cat([s4, torch.from_numpy(PX.uniform_fill(7, (36, 10), 93, -1, 1)).to(m.device)], 0)
    Gb = m.calculate_G(big, m.pi_one_hot.repeat(10, 1), samples=10, stage=4)
    assert torch.equal(Gb[0][:4], res[1][1][0]) and torch.equal(Gb[4][:4], res[1][1][4])


def test_rows_are_an_argument_of_the_call_not_context_state(models):
    """ABI 4 (efe_rows): the liveness mask and the row identities belong to ONE call.  A masked call leaves the next plain call on the same
    context untouched (with the efe_set_row_mask shim of ABI 2 - 5 it saw the mask until someone cleared it); a COMPACTED call -- only the live entries, as a
    dense batch with their ids -- returns bit for bit the rows of the full batch (noise keys follow the entry id, not the slot), for
    calculate_G, calculate_G_mean and simulate_batch, with device noise and with injected noise."""
    from daimc_amd.model import Rows
    m = models(1234, 1.15, 33)
    A, Eps, T = 4, 7, 3
    M = A * Eps
    s0 = torch.from_numpy(PX.uniform_fill(2, (M, 10), 77, -1, 1)).to(m.device)
    pi0 = torch.from_numpy(np.eye(4, dtype=np.float32)[np.arange(M) % 4]).to(m.device)
    starts = torch.from_numpy(PX.uniform_fill(9, (Eps, 10), 78, -1, 1)).to(m.device)
    alive = torch.tensor([1, 0, 1, 1, 0, 0, 1], dtype=torch.uint8, device=m.device)
    keep = torch.nonzero(alive).flatten()
    ids = keep.to(torch.int32)
    krows = (keep[:, None] * A + torch.arange(A, device=m.device)[None]).reshape(-1)
    for inj in (False, True):
        m.eps_source, m.u_source = (PX.normals, PX.uniforms) if inj else (None, None)
        ref = m.calculate_G(s0, pi0, samples=3, stage=4)
        refm = m.calculate_G_mean(s0, pi0, stage=5)
        rsim = m.simulate_batch(starts, T, use_means=False, stage=9)
        # mask as an argument; the next plain call is complete
        out = m.calculate_G(s0, pi0, samples=3, stage=4, rows=Rows(mask=alive, rows_per_entry=A))
        assert torch.equal(out[0][krows], ref[0][krows]) and torch.equal(out[4][krows], ref[4][krows])
        again = m.calculate_G(s0, pi0, samples=3, stage=4)
        assert torch.equal(again[0], ref[0]) and torch.equal(again[2], ref[2])
        # compacted calls
        rc = Rows(ids=ids, rows_per_entry=A, ids_host=keep.tolist())
        cmp_ = m.calculate_G(s0[krows], pi0[krows], samples=3, stage=4, rows=rc)
        assert torch.equal(cmp_[0], ref[0][krows])
        for k in range(3):
            assert torch.equal(cmp_[1][k], ref[1][k][krows])
        assert torch.equal(cmp_[2], ref[2][krows]) and torch.equal(cmp_[3], ref[3][krows]) and torch.equal(cmp_[4], ref[4][krows])
        cmpm = m.calculate_G_mean(s0[krows], pi0[krows], stage=5, rows=rc)
        assert torch.equal(cmpm[0], refm[0][krows]) and torch.equal(cmpm[2], refm[2][krows])
        csim = m.simulate_batch(starts[keep], T, use_means=False, stage=9, rows=Rows(ids=ids, ids_host=keep.tolist()))
        assert torch.equal(csim[0], rsim[0][keep]) and torch.equal(csim[1], rsim[1][keep]) and torch.equal(csim[2], rsim[2][keep])
        # compacted AND masked: the mask is read at the entry id
        alive2 = alive.clone(); alive2[2] = 0
        cm = m.calculate_G(s0[krows], pi0[krows], samples=3, stage=4, rows=Rows(mask=alive2, ids=ids, rows_per_entry=A, ids_host=keep.tolist()))
        live2 = (alive2[keep] != 0).repeat_interleave(A)
        assert torch.equal(cm[0][live2], ref[0][krows][live2])
    m.eps_source, m.u_source = None, None
    with pytest.raises(ValueError):
        Rows(mask=torch.ones(5, dtype=torch.float32, device=m.device))
    with pytest.raises(RuntimeError):
        m.calculate_G(s0[:6], pi0[:6], samples=1, stage=0, rows=Rows(mask=alive, rows_per_entry=A))      # 6 rows are not whole entries


def test_batched_mcts_compaction_changes_nothing(models):
    """the lock-step planner gathering the live episodes into a dense batch as they stop (efe_rows.ids) == masking them == evaluating
    every episode: identical paths, iteration counts, G histories and visit distributions -- with the simulations on the second stream"""
    import daimc_amd
    m = models(1234, 1.15, 21)
    frames = torch.from_numpy(synth.make_frames(56, 24)[:, 0][:, None])
    found = False
    for thr in (0.3, 0.25, 0.2, 0.15):
        res, compacted = [], []
        for skip, comp in ((True, True), (True, False), (False, False)):
            p = daimc_amd.MCTS_Params()
            p.repeats, p.simulation_depth, p.use_means, p.samples, p.threshold = 26, 3, False, 2, thr
            p.skip_stopped, p.compact_stopped = skip, comp
            m._stage = 0
            res.append(daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64)))
            pl = [q for q in m._planners.values() if q.E == 24 and q.p.compact_stopped == comp and q.p.skip_stopped == skip and q.p.threshold == thr][-1]
            compacted.append(pl._ids is not None and len(pl._ids[1]) < 24)
        for (out_b, dist_b) in res[1:]:
            for e in range(24):
                assert res[0][0][e][0] == out_b[e][0] and res[0][0][e][1] == out_b[e][1] and res[0][0][e][3] == out_b[e][3] and res[0][0][e][4] == out_b[e][4]
            assert torch.equal(res[0][1], dist_b)
        assert not compacted[1] and not compacted[2]
        stops = [o[1] for o in res[0][0]]
        if compacted[0] and max(stops) == 26:          # some episodes were compacted away while others ran to the end
            found = True
            break
    assert found, 'no threshold made the planner compact its batch: the fixture does not exercise efe_rows.ids'


def test_cached_simulation_replica_follows_the_weights(models, weights_cache):
    """the lock-step planner's simulation replica is cached on the model per weight version: after new weights are loaded the
    next decision must plan with them (== a fresh model with those weights), not with the cached copy"""
    import daimc_amd
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.use_means, p.threshold, p.samples = 4, 3, False, 0.9, 2
    frames = torch.from_numpy(synth.make_frames(57, 8)[:, 0][:, None])           # 8 episodes: the overlapped (replica) path
    m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=3, init_weights=False)
    m.load_flat_weights(weights_cache(1234, 1.0))
    m._stage = 0
    daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
    r1 = m._replica
    m._stage = 0
    daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
    assert m._replica is r1                                                      # reused
    m.load_flat_weights(weights_cache(1234, 1.35))
    m._stage = 0
    out, dist = daimc_amd.active_inference_mcts_batch(m, frames, p, o_shape=(1, 64, 64))
    assert m._replica is not r1
    fresh = models(1234, 1.35, 3)
    fresh._stage = 0
    out2, dist2 = daimc_amd.active_inference_mcts_batch(fresh, frames, p, o_shape=(1, 64, 64))
    for e in range(8):
        assert out[e][0] == out2[e][0] and out[e][3] == out2[e][3] and out[e][4] == out2[e][4]
    assert torch.equal(dist, dist2)


def test_contexts_release_their_device_memory(weights_cache):
    """efe_destroy (model deletion) returns the context's packed weights and scratch arena: creating, using and dropping
    contexts in a loop must not grow the device memory in use"""
    import gc
    import daimc_amd
    w = weights_cache(1234, 1.15)
    s = PX.uniform_fill(2, (8, 10), 5, -1, 1)
    pi = np.eye(4, dtype=np.float32)[np.arange(8) % 4]

    def once():
        m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=1, init_weights=False)
        m.load_flat_weights(w)
        g = m.calculate_G(s, pi, samples=2, stage=0)[0]
        torch.cuda.synchronize()
        del m
        gc.collect()
        return g

    once()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(6):
        once()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, f'{(free0 - free1) >> 20} MiB of device memory not returned after 6 create/destroy cycles'


# ---------------------------------------------------------------------------------------------------------------------------
# OPT-IN EXPERIMENTS (engine options mfma_bf16x3 / mfma_f16x2, csrc/bf16x3.hip): the decoder's Linear(256, 16384) and its three large
# transposed convolutions on the 16-bit matrix pipe with BOTH operands split -- into three bf16 planes (6 products) or two fp16 planes
# (3 products, weights scaled by a power of two), fp32 accumulation.  Narrower inputs than the reference's fp32 -- never the default,
# never the headline -- so every decoder-bearing fixture is run through both at the UNCHANGED tolerances
# ---------------------------------------------------------------------------------------------------------------------------
SPLITS = ['mfma_bf16x3', 'mfma_f16x2']


@pytest.fixture(params=SPLITS)
def models_b3(request, models):
    used = []

    def get(wseed, gain, seed):
        m = models(wseed, gain, seed)
        m.set_option(request.param, 1)
        used.append(m)
        return m
    yield get
    for m in used:
        m.set_option(request.param, 0)


def test_split_operands_every_decoder_fixture_at_unchanged_tolerances(golden, models_b3):
    """the reference-captured fixtures that go through the decoder, with the experiment on: same assertions, same tolerances"""
    for gain in GAINS:
        test_networks_vs_golden(golden, models_b3, gain)
        for case in ('m4s1', 'm6s3'):
            test_calculate_G_vs_golden(golden, models_b3, gain, case)
        test_calculate_G_mean_vs_golden(golden, models_b3, gain)
    for name in ('rollout_cfg1', 'rollout_m8d2s2', 'rollout_m8d2s2mean'):
        test_rollout_vs_golden(golden, models_b3, name)
    for name in ('rollout4_s2', 'rollout4_mean'):
        test_rollout4_vs_golden(golden, models_b3, name)
    for name in ('simulate_sample', 'simulate_means'):
        test_simulate_vs_golden(golden, models_b3, name)
    for name in ('mcts_deep_s10_thr',):
        test_planners_at_benchmark_depth_vs_reference(golden, models_b3, name)
    # ... and the reference planner's own default call (300 repeats, use_means: 480-image expansions of the 40-episode batch) and two
    # simulations per iteration -- every path, stop and visit count as captured
    for name in ('mcts_defaults', 'mcts_simrep2_s10'):
        test_planner_at_the_reference_defaults_and_with_two_simulations(golden, models_b3, name)


def test_split_operands_large_launches_vs_oracle(models_b3, weights_cache):
    """the launch sizes at which k_dec_a_b3 / k_dec_b_b3 run (the persistent kernels: more than 128 images per launch), against the oracle at
    the unchanged tolerances: 1100 decoder rows, calculate_G over 150 rows x 4 samples, and the benchmarked configuration itself (128 rows x
    depth 5 x 10 samples)"""
    test_decoder_many_rows_vs_oracle(models_b3, weights_cache)
    test_calculate_G_many_rows_vs_oracle(models_b3, weights_cache)
    test_full_size_cfg2_vs_oracle(models_b3, weights_cache)


@pytest.mark.parametrize('opt', SPLITS)
def test_split_operands_launch_groups_rows_and_masks_change_nothing(models, opt):
    """inside ONE split mode results are a function of the noise keys alone as long as every launch group stays on the persistent split
    kernels (> 128 images): decoder launch groups of 360 / 240 images instead of one of 720, a global row offset, and a liveness mask (live rows
    bit-identical; k_dec_a_b3 / k_dec_b_b3 skip dead images, the persistent image loop walks on to the next live one)"""
    from daimc_amd.model import Rows
    m = models(1234, 1.15, 12)
    M, S = 60, 4                                        # 720 decoder images per call
    s0 = torch.from_numpy(PX.uniform_fill(2, (M, 10), 91, -1, 1)).to(m.device)
    pi0 = torch.from_numpy(np.eye(4, dtype=np.float32)[np.arange(M) % 4]).to(m.device)
    try:
        m.set_option(opt, 1)
        ref = m.calculate_G(s0, pi0, samples=S, stage=4)
        for chunk in (360, 240):             # (720 = 2 x 360 = 3 x 240: every group stays above the 128-image small-launch threshold)
            m.set_option('dec_chunk', chunk)
            got = m.calculate_G(s0, pi0, samples=S, stage=4)
            assert all(torch.equal(a_, b_) for a_, b_ in zip((got[0], got[2], got[4]), (ref[0], ref[2], ref[4]))), chunk
        m.set_option('dec_chunk', 32768)
        # rows 8 .. 59 evaluated alone at their global offset
        part = m.calculate_G(s0[8:], pi0[8:], samples=S, stage=4, row_offset=8)
        assert torch.equal(part[0], ref[0][8:]) and torch.equal(part[4], ref[4][8:])
        alive = torch.ones(M // 4, dtype=torch.uint8, device=m.device); alive[[0, 3, 4, 14]] = 0
        out = m.calculate_G(s0, pi0, samples=S, stage=4, rows=Rows(mask=alive, rows_per_entry=4))
        rows = alive.bool().repeat_interleave(4)
        assert torch.equal(out[0][rows], ref[0][rows]) and torch.equal(out[4][rows], ref[4][rows])
    finally:
        m.set_option('dec_chunk', 32768)
        m.set_option(opt, 0)


@pytest.mark.parametrize('opt', SPLITS)
def test_split_operands_vs_fp32_path_and_oracle_on_many_rows(models, weights_cache, opt):
    """1100 decoder rows (several 64-row tiles + a ragged tail, every feature group): the experiment against the fp32 kernels (same masks,
    images within the sigmoid tolerance) and against the oracle; and it really is another kernel (not bit-identical); switching from one
    split to the other re-packs the planes"""
    seed, M, st = 31, 1100, 5
    m = models(1234, 1.15, seed)
    s = PX.uniform_fill(12, (M, 10), 400, -1.5, 1.5)
    ref = c(m.model_down.decoder(s, stage=st, pass_=PX.PASS_D2A))
    other = SPLITS[1 - SPLITS.index(opt)]
    try:
        m.set_option(other, 1)
        got_other = c(m.model_down.decoder(s, stage=st, pass_=PX.PASS_D2A))
        m.set_option(opt, 1)                       # the planes of the other mode are released, these packed
        got = c(m.model_down.decoder(s, stage=st, pass_=PX.PASS_D2A))
    finally:
        m.set_option(opt, 0)
    assert np.array_equal(c(m.model_down.decoder(s, stage=st, pass_=PX.PASS_D2A)), ref)        # off again: the exact fp32 kernels, bit for bit
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=4e-6)
    assert not np.array_equal(got, ref) and not np.array_equal(got, got_other)
    orc = EO.OracleModel(weights_cache(1234, 1.15), EO.PhiloxNoise(seed))
    with torch.no_grad():
        o = orc.decoder(torch.from_numpy(s[:200]), PX.PASS_D2A, 0, st).numpy()
    np.testing.assert_allclose(got[:200], o, rtol=1e-5, atol=4e-6)


def test_fp16_split_weight_scale_follows_the_weights(models, weights_cache):
    """mfma_f16x2 scales every layer's weights by a power of two before the fp16 split (largest magnitude just below 2^14) and multiplies the
    accumulators back: results must not depend on the overall magnitude of a layer beyond fp32 rounding -- the decoder with ConvT3's weights
    x 2^-9 and the final layer's x 2^9 (y3 a factor 512 smaller: every low plane would be a denormal without the scale) stays within the
    sigmoid tolerance of the exact fp32 kernels on the same weights"""
    import daimc_amd
    w = dict(weights_cache(1234, 1.15))
    w['down.po_net.17.weight'] = w['down.po_net.17.weight'] * np.float32(2.0 ** -9)
    w['down.po_net.17.bias'] = w['down.po_net.17.bias'] * np.float32(2.0 ** -9)
    w['down.po_net.19.weight'] = w['down.po_net.19.weight'] * np.float32(2.0 ** 9)
    m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=5, init_weights=False)
    m.load_flat_weights(w)
    s = PX.uniform_fill(13, (300, 10), 401, -1.5, 1.5)
    ref = c(m.model_down.decoder(s, stage=2, pass_=PX.PASS_D1))
    m.set_option('mfma_f16x2', 1)
    got = c(m.model_down.decoder(s, stage=2, pass_=PX.PASS_D1))
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=4e-6)
    base = models(1234, 1.15, 5)
    np.testing.assert_allclose(ref, c(base.model_down.decoder(s, stage=2, pass_=PX.PASS_D1)), rtol=1e-5, atol=4e-6)     # (the rescaled network IS the same function)


def test_fp16_split_overflow_is_loud(weights_cache):
    """the stated limit of mfma_f16x2: an activation beyond fp16's range (65 504) cannot be split -- the affected images come out NaN
    (hi = inf, lo = x - inf), never as finite wrong numbers; the exact fp32 kernels and the bf16 split (fp32's exponent range) are unaffected"""
    import daimc_amd
    w = dict(weights_cache(1234, 1.15))
    w['down.po_net.9.weight'] = w['down.po_net.9.weight'] * np.float32(1.0e6)       # x4 = relu(W h + b) * 2 reaches ~1e6
    w['down.po_net.13.weight'] = w['down.po_net.13.weight'] * np.float32(1.0e-6)
    m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=5, init_weights=False)
    m.load_flat_weights(w)
    s = PX.uniform_fill(13, (300, 10), 401, -1.5, 1.5)
    ref = c(m.model_down.decoder(s, stage=2, pass_=PX.PASS_D1))
    assert np.isfinite(ref).all()
    m.set_option('mfma_bf16x3', 1)
    np.testing.assert_allclose(c(m.model_down.decoder(s, stage=2, pass_=PX.PASS_D1)), ref, rtol=1e-5, atol=4e-6)
    m.set_option('mfma_f16x2', 1)
    got = c(m.model_down.decoder(s, stage=2, pass_=PX.PASS_D1))
    bad = ~np.isfinite(got).all(axis=(1, 2, 3))
    assert bad.any()                                                   # some image saw an activation above 65 504 ...
    np.testing.assert_allclose(got[~bad], ref[~bad], rtol=1e-5, atol=4e-6)      # ... and every image that did not is still right


def test_split_operands_are_off_by_default_and_refused_on_other_geometries(models):
    import daimc_amd
    m = models(1234, 1.15, 3)
    assert all(getattr(m, '_opts', {}).get(o, 0) == 0 for o in SPLITS)
    g = daimc_amd.ActiveInferenceModel(10, 3, 0.0, 1.0, 1.0, colour_channels=3, resolution=32, device='cuda:0', seed=1)
    for o in SPLITS:
        with pytest.raises(RuntimeError):
            g.set_option(o, 1)
