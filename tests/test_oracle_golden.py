"""CPU tests: the oracle (oracle/efe_oracle.py) against the fixtures captured from the shimmed reference
(oracle/make_golden.py), plus known-answer tests of the Philox generator."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import philox as PX
from oracle import synth
from oracle import efe_oracle as EO

GAINS = ['g100', 'g115', 'g135']


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert [int(x) for x in PX.philox4x32_10(0, 0, 0, 0, 0, 0)] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert [int(x) for x in PX.philox4x32_10(*([0xffffffff] * 4), 0xffffffff, 0xffffffff)] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert [int(x) for x in PX.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)] == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_noise_addressable():
    a = PX.dropout_mask(5, PX.TAG_DEC + 3, 8, 16384, PX.PASS_D1, 2, 9, row_offset=0)
    b = PX.dropout_mask(5, PX.TAG_DEC + 3, 4, 16384, PX.PASS_D1, 2, 9, row_offset=4)
    assert np.array_equal(a[4:], b)                       # rows are keyed globally (multi-GPU invariance)
    assert set(np.unique(a)) == {0.0, 2.0}
    assert abs(a.mean() - 1.0) < 0.01
    z = PX.normals(5, 4096, 10, PX.PASS_T1, 0, 0)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    assert np.array_equal(PX.normals(5, 8, 10, 0, 0, 0)[3:], PX.normals(5, 5, 10, 0, 0, 0, row_offset=3))


def test_helpers_vs_reference(golden):
    g = golden('helpers')
    p = torch.from_numpy(g['p'])
    assert np.array_equal(EO.entropy_normal_from_logvar(torch.from_numpy(g['lv'])).numpy(), g['ent_normal'])
    assert np.array_equal(EO.entropy_bernoulli(p).numpy(), g['ent_bern'])
    assert np.array_equal(EO.check_reward(p).numpy(), g['reward'])
    # restated closed form of the NCHW-broadcast reward
    d1, d0 = np.float32(1.00001), np.float32(1e-5)
    o = g['p'][:, 0]
    top = o * np.log(d1) + (1 - o) * np.log(np.float32(d1 - np.float32(1)))
    bot = o * np.log(d0) + (1 - o) * np.log(d1)
    h = np.arange(64)[None, :, None]
    r = np.where(h < 32, top, bot).reshape(len(o), -1).mean(1) * 10
    np.testing.assert_allclose(r, g['reward'], rtol=2e-6)


def test_softmax_multi_with_log(golden):
    g = golden('rollout_m8d2s2')
    P, logP = EO.softmax_multi_with_log(-g['sum_G'], 4)
    np.testing.assert_allclose(P, g['Ppi'], rtol=1e-6)
    np.testing.assert_allclose(logP, g['logPpi'], rtol=1e-6, atol=1e-6)


def _oracle(g, weights_cache):
    w = weights_cache(g['wseed'], g['gain'])
    return EO.OracleModel(w, EO.PhiloxNoise(int(g['nseed'])))


@pytest.mark.parametrize('gain', GAINS)
def test_networks_vs_reference(golden, weights_cache, gain):
    g = golden(f'nets_{gain}')
    m = _oracle(g, weights_cache)
    st = int(g['stage'])
    with torch.no_grad():
        ps1, mean, lv = m.transition_with_sample(torch.from_numpy(g['pi']), torch.from_numpy(g['s']), PX.PASS_T1, 0, st)
        po = m.decoder(torch.from_numpy(g['s']), PX.PASS_D1, 0, st)
        es, em, elv = m.encoder_with_sample(torch.from_numpy(g['frames']), PX.PASS_E1, 0, st)
        hl, hq, hlq = m.encode_s(torch.from_numpy(g['s']))
    for a, b in ((ps1, 't_ps1'), (mean, 't_mean'), (lv, 't_lv'), (po, 'd_po'), (es, 'e_s'), (em, 'e_mean'), (elv, 'e_lv'),
                 (hl, 'h_logits'), (hq, 'h_q'), (hlq, 'h_logq')):
        np.testing.assert_allclose(a.numpy(), g[b], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('gain', GAINS)
@pytest.mark.parametrize('case', ['m4s1', 'm6s3'])
def test_calculate_G_vs_reference(golden, weights_cache, gain, case):
    g = golden(f'calcG_{case}_{gain}')
    m = _oracle(g, weights_cache)
    with torch.no_grad():
        G, terms, ps1, ps1m, po1 = m.calculate_G(torch.from_numpy(g['s0']), torch.from_numpy(g['pi0']), int(g['samples']), int(g['stage']))
    np.testing.assert_allclose(G.numpy(), g['G'], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(ps1.numpy(), g['ps1'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(po1.numpy(), g['po1'], rtol=1e-6, atol=1e-6)


def test_rollouts_vs_reference(golden, weights_cache):
    for name, four in (('rollout_cfg1', False), ('rollout_m8d2s2', False), ('rollout_m8d2s2mean', False),
                       ('rollout4_s2', True), ('rollout4_mean', True)):
        g = golden(name)
        m = _oracle(g, weights_cache)
        with torch.no_grad():
            if four:
                G, T, po1 = m.calculate_G_4_repeated(torch.from_numpy(g['o']), int(g['steps']), bool(g['calc_mean']), int(g['samples']), int(g['stage']))
            else:
                G, T, po1 = m.calculate_G_repeated(torch.from_numpy(g['o']), torch.from_numpy(g['pi']), int(g['steps']),
                                                   bool(g['calc_mean']), int(g['samples']), int(g['stage']))
        np.testing.assert_allclose(G.numpy(), g['sum_G'], rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(po1.numpy(), g['po1'], rtol=1e-6, atol=1e-6)


def test_simulate_vs_reference(golden, weights_cache):
    for name in ('simulate_sample', 'simulate_means'):
        g = golden(name)
        m = _oracle(g, weights_cache)
        with torch.no_grad():
            G, pi0, q = m.mcts_step_simulate(torch.from_numpy(g['start']), int(g['depth']), bool(g['use_means']), int(g['stage']),
                                             episode=int(g['episode']))
        assert abs(G - float(g['G'])) < 1e-3
        assert np.array_equal(pi0.numpy(), g['pi0'])
        np.testing.assert_allclose(q.numpy(), g['Qpi'], rtol=1e-6)


def test_simulate_invalid_posterior_vs_reference(golden, weights_cache):
    """the bare-except fallback of mcts_step_simulate (/root/reference/src/torchmodel.py:362-367, 378-381), captured from the reference
    with a habit network whose posterior is NaN / [0, 0, NaN, 0] (oracle/make_golden_invalid.py): action 0 on every step, Qpi = one-hot"""
    for name in ('simulate_invalid_nan', 'simulate_invalid_inf'):
        g = golden(name)
        w = dict(weights_cache(int(g['wseed']), float(g['gain'])))
        b = np.array(w['top.qpi_net.4.bias'], copy=True)
        b[int(g['bias_index'])] = g['bias_value']
        w['top.qpi_net.4.bias'] = b
        m = EO.OracleModel(w, EO.PhiloxNoise(int(g["nseed"])))
        with torch.no_grad():
            G, pi0, q = m.mcts_step_simulate(torch.from_numpy(g['start']), int(g['depth']), False, int(g['stage']), episode=int(g['episode']))
        assert abs(G - float(g['G'])) < 1e-3
        assert np.array_equal(pi0.numpy(), g['pi0']) and np.array_equal(pi0.numpy(), np.eye(4, dtype=np.float32)[[0] * int(g['depth'])])
        assert np.array_equal(q.numpy(), g['Qpi'])


def test_convtranspose_subpixel_restatement():
    """independent numpy restatement of ConvTranspose2d(k3,s2,p1,op1) in the 4-parity form the HIP kernel
    uses against torch's conv_transpose2d."""
    rng = np.random.default_rng(0)
    n, ci, co = 5, 3, 4
    x = rng.standard_normal((1, ci, n, n)).astype(np.float32)
    w = rng.standard_normal((ci, co, 3, 3)).astype(np.float32)
    ref = torch.nn.functional.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(w), stride=2, padding=1, output_padding=1).numpy()
    out = np.zeros((1, co, 2 * n, 2 * n), dtype=np.float64)
    xp = np.zeros((ci, n + 1, n + 1)); xp[:, :n, :n] = x[0]
    for ph in (0, 1):
        for pw in (0, 1):
            khs = [(1, 0)] if ph == 0 else [(0, 1), (2, 0)]
            kws = [(1, 0)] if pw == 0 else [(0, 1), (2, 0)]
            for kh, da in khs:
                for kw, db in kws:
                    src = xp[:, da:da + n, db:db + n]                      # in[a+da, b+db]
                    out[0, :, ph::2, pw::2] += np.einsum('iab,io->oab', src, w[:, :, kh, kw])
    np.testing.assert_allclose(out, ref, atol=1e-5)


# ------------------------------------------------------------------------------------------------------
# the oracle planner (oracle/mcts_oracle.py) against the fixtures captured from the reference planner
# ------------------------------------------------------------------------------------------------------
def _paths(arr):
    return [[int(a) for a in row if a >= 0] for row in arr]


@pytest.mark.parametrize('name', ['mcts_means', 'mcts_samples', 'mcts_prior'])
def test_oracle_planner_vs_reference(golden, weights_cache, name):
    from oracle import mcts_oracle as MO
    g = golden(name)
    m = _oracle(g, weights_cache)
    p = MO.Params(repeats=int(g['repeats']), simulation_depth=int(g['simulation_depth']), use_means=bool(g['use_means']),
                  threshold=float(g['threshold']), using_prior_for_exploration=(name == 'mcts_prior'))
    path, reps, explored, all_paths, all_G, _ = MO.plan(m, torch.from_numpy(g['frame']), p, int(g['stage']))
    assert reps == int(g['repeats_done']) and explored == int(g['states_explored'])
    assert all_paths == _paths(g['all_paths'])
    np.testing.assert_allclose(np.array(all_G), g['all_paths_G'], atol=1e-3)
    assert path == [int(a) for a in g['final_path']]


@pytest.mark.parametrize('e', [1, 6])
def test_oracle_planner_ten_samples_vs_reference(golden, weights_cache, e):
    """BASELINE configs[2] shape (Node.expand(samples=10), simulation depth 5), episode e of the 8-episode fixture
    (episode 1 stops early, episode 6 runs all repeats)"""
    from oracle import mcts_oracle as MO
    g = golden('mcts_batch_s10')
    m = _oracle(g, weights_cache)
    p = MO.Params(repeats=int(g['repeats']), simulation_depth=int(g['simulation_depth']), use_means=False,
                  threshold=float(g['threshold']), samples=int(g['samples']))
    path, reps, explored, all_paths, all_G, root_N = MO.plan(m, torch.from_numpy(g['frames'][e]), p, int(g['stage']), episode=e)
    n = int(g['n_paths'][e])
    assert reps == int(g['repeats_done'][e]) and explored == int(g['states_explored'][e]) and len(all_paths) == n
    assert all_paths == _paths(g['all_paths'][e][:n])
    np.testing.assert_allclose(np.array(all_G), g['all_paths_G'][e][:n], atol=1e-3)
    assert path == [int(a) for a in g['final_path'][e] if a >= 0]
    assert np.array_equal(root_N.numpy(), g['root_N'][e])


def _check_planner_fixture(g, e, got):
    path, reps, explored, all_paths, all_G, root_N = got
    n = int(g['n_paths'][e])
    assert reps == int(g['repeats_done'][e]) and explored == int(g['states_explored'][e]) and len(all_paths) == n
    assert all_paths == _paths(g['all_paths'][e][:n])
    np.testing.assert_allclose(np.array(all_G), g['all_paths_G'][e][:n], atol=1e-3)
    assert path == [int(a) for a in g['final_path'][e] if a >= 0]
    assert np.array_equal(np.asarray(root_N), g['root_N'][e])


def test_oracle_planner_at_benchmark_depth_vs_reference(golden, weights_cache):
    """BASELINE configs[2] at its real depth: 50 iterations x Node.expand(samples=10) x depth-5 simulations, early stop disabled
    (205 nodes, paths up to 6 actions) -- one of the three episodes captured from the reference planner (the GPU suite runs all three)"""
    from oracle import mcts_oracle as MO
    g = golden('mcts_deep_s10')
    m = _oracle(g, weights_cache)
    p = MO.Params(repeats=int(g['repeats']), simulation_depth=int(g['simulation_depth']), use_means=False,
                  threshold=float(g['threshold']), samples=int(g['samples']))
    e = 1
    assert int(g['n_nodes'][e]) == 1 + 4 * 51 and int(g['repeats_done'][e]) == 50
    got = MO.plan(m, torch.from_numpy(g['frames'][e]), p, int(g['stage']), episode=e)
    _check_planner_fixture(g, e, (got[0], got[1], got[2], got[3], got[4], got[5].numpy()))


@pytest.mark.parametrize('name,e', [('mcts_deep_s10_thr', 4), ('mcts_deep_s10_thr04', 2)])
def test_oracle_planner_early_stop_at_benchmark_depth_vs_reference(golden, weights_cache, name, e):
    """the reference planner's early stop (mcts.py:170-181) at the benchmark's depth: threshold 0.5 (the reference's default, bench.py's
    early-stop leg) stops episode 4 before iteration 29, threshold 0.4 stops episode 2 before 19 (oracle/make_golden_thr.py)"""
    from oracle import mcts_oracle as MO
    g = golden(name)
    m = _oracle(g, weights_cache)
    p = MO.Params(repeats=int(g['repeats']), simulation_depth=int(g['simulation_depth']), use_means=False,
                  threshold=float(g['threshold']), samples=int(g['samples']))
    assert 10 <= int(g['repeats_done'][e]) <= 45
    # the stop statistic of the last check exceeded the threshold, every earlier one did not
    ts = g['thr_stat'][e][:int(g['repeats_done'][e]) + 1]
    assert ts[-1] > float(g['threshold']) and (ts[:-1] <= float(g['threshold'])).all()
    got = MO.plan(m, torch.from_numpy(g['frames'][e]), p, int(g['stage']), episode=e)
    _check_planner_fixture(g, e, (got[0], got[1], got[2], got[3], got[4], got[5].numpy()))


def _params_of(g, MO):
    return MO.Params(repeats=int(g['repeats']), simulation_depth=int(g['simulation_depth']), simulation_repeats=int(g['simulation_repeats']),
                     use_means=bool(g['use_means']), threshold=float(g['threshold']), C=float(g['C']), samples=int(g['samples']))


@pytest.mark.parametrize('name,e', [('mcts_defaults', 2), ('mcts_defaults', 5), ('mcts_defaults_full', 1)])
def test_oracle_planner_at_the_reference_defaults(golden, weights_cache, name, e):
    """the reference planner with MCTS_Params() UNTOUCHED (mcts.py:139-148: 300 repeats, depth 3, use_means -> calculate_G_mean expansions,
    threshold 0.5; oracle/make_golden_defaults.py): fixture episode 2 stops before iteration 177, episode 5 before 21; with the stop out of
    reach (mcts_defaults_full) every episode grows the full 1 + 4 * 301 = 1 205-node tree.  Fixture episode k is GLOBAL episode
    episode_ids[k] (its noise rows).  The GPU suite runs every episode."""
    from oracle import mcts_oracle as MO
    g = golden(name)
    m = _oracle(g, weights_cache)
    p = _params_of(g, MO)
    d = MO.Params()
    # the selection was made for margins: no tree-policy decision of these episodes is closer than 5e-5 (fp32 differences between two
    # implementations of G move a score by ~1e-5 at most)
    n = g['n_paths']
    assert min(float(g['sel_margin'][k, :n[k]].min()) for k in range(int(g['episodes']))) >= 5e-5
    assert np.array_equal(synth_frames(g)[g['episode_ids']], g['frames'])
    if name == 'mcts_defaults':          # the fixture's parameters ARE the defaults
        assert all(getattr(p, k) == getattr(d, k) for k in ('C', 'threshold', 'repeats', 'simulation_repeats', 'simulation_depth', 'use_means', 'samples'))
        assert [int(x) for x in g['repeats_done']] == [300, 217, 177, 125, 65, 21]
        r = int(g['repeats_done'][e])
        ts = g['thr_stat'][e][:r + 1]
        assert ts[-1] > 0.5 and (ts[:-1] <= 0.5).all()
    else:
        assert (g['repeats_done'] == 300).all() and (g['n_nodes'] == 1 + 4 * 301).all()
    got = MO.plan(m, torch.from_numpy(g['frames'][e]), p, int(g['stage']), episode=int(g['episode_ids'][e]))
    _check_planner_fixture(g, e, (got[0], got[1], got[2], got[3], got[4], got[5].numpy()))


def synth_frames(g):
    from oracle import synth
    return synth.make_frames(int(g['frame_seed']), int(g['n_frames']))


@pytest.mark.parametrize('e', [0, 3])
def test_oracle_planner_two_simulations_per_iteration(golden, weights_cache, e):
    """simulation_repeats = 2 (mcts.py:185-189: the mean of the simulations is back-propagated, the leaf keeps the LAST simulation's Qpi) at
    the benchmark's depth; episode 3 stops before iteration 45"""
    from oracle import mcts_oracle as MO
    g = golden('mcts_simrep2_s10')
    m = _oracle(g, weights_cache)
    p = _params_of(g, MO)
    assert p.simulation_repeats == 2 and [int(x) for x in g['repeats_done']] == [50, 50, 50, 45]
    assert (g['states_explored'] == g['n_paths'] * p.simulation_depth * 2).all()
    assert [int(x) for x in g['episode_ids']] == [0, 1, 2, 3]
    got = MO.plan(m, torch.from_numpy(g['frames'][e]), p, int(g['stage']), episode=e)
    _check_planner_fixture(g, e, (got[0], got[1], got[2], got[3], got[4], got[5].numpy()))


@pytest.mark.parametrize('e', [0, 1])
def test_oracle_planner_prior_with_ten_samples_vs_reference(golden, weights_cache, e):
    """using_prior_for_exploration (mcts.py:44-45) together with Node.expand(samples=10) and use_habit (shortcut evaluated, not taken)"""
    from oracle import mcts_oracle as MO
    g = golden('mcts_prior_s10')
    m = _oracle(g, weights_cache)
    p = MO.Params(repeats=int(g['repeats']), simulation_depth=int(g['simulation_depth']), use_means=False, threshold=float(g['threshold']),
                  samples=int(g['samples']), using_prior_for_exploration=True, use_habit=True)
    got = MO.plan(m, torch.from_numpy(g['frames'][e]), p, int(g['stage']), episode=e)
    _check_planner_fixture(g, e, (got[0], got[1], got[2], got[3], got[4], got[5].numpy()))


def test_resolution32_networks_vs_reference(golden):
    """the reference's own Animal-AI-branch model (pi 3, 3 x 32 x 32, last_strides = 1, torchmodel.py:77-80): its networks run (its
    calculate_G does not: calc_reward_animalai is undefined), so the geometry-generic restatement is pinned at network level"""
    g = golden('nets32_g115')
    A, C, R = int(g['pi_dim']), int(g['channels']), int(g['resolution'])
    m = EO.OracleModel(synth.make_weights(int(g['wseed']), float(g['gain']), A, C, R), EO.PhiloxNoise(int(g['nseed'])),
                       pi_dim=A, channels=C, resolution=R)
    st = int(g['stage'])
    with torch.no_grad():
        ps1, mean, lv = m.transition_with_sample(torch.from_numpy(g['pi']), torch.from_numpy(g['s']), PX.PASS_T1, 0, st)
        po = m.decoder(torch.from_numpy(g['s']), PX.PASS_D1, 0, st)
        es, em, elv = m.encoder_with_sample(torch.from_numpy(g['frames']), PX.PASS_E1, 0, st)
        hl, hq, hlq = m.encode_s(torch.from_numpy(g['s']))
    assert po.shape == (5, 3, 32, 32) and hq.shape == (5, 3)
    for a, b in ((ps1, 't_ps1'), (mean, 't_mean'), (lv, 't_lv'), (po, 'd_po'), (es, 'e_s'), (em, 'e_mean'), (elv, 'e_lv'),
                 (hl, 'h_logits'), (hq, 'h_q'), (hlq, 'h_logq')):
        np.testing.assert_allclose(a.numpy(), g[b], rtol=1e-6, atol=1e-6)


def test_upstream_intent_reward_vs_reference_formula(golden):
    """oracle.efe_oracle.check_reward_upstream_intent (the engine option reward_upstream_intent) == the reference's own calc_reward /
    check_reward formula applied to the NHWC view of the batch (captured by oracle/make_golden_deep.py) -- SURVEY appendix C"""
    g = golden('helpers_intent')
    got = EO.check_reward_upstream_intent(torch.from_numpy(g['p']))
    np.testing.assert_allclose(got.numpy(), g['reward_upstream_intent'], rtol=1e-6)
    assert not np.allclose(EO.check_reward(torch.from_numpy(g['p'])).numpy(), g['reward_upstream_intent'], rtol=1e-3)
