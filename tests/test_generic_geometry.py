"""SURVEY row a-13 / BASELINE configs[4] (-m gpu): the geometry-generic path of the engine (pi_dim 3, 3 x 84 x 84 observations;
generic.hip) against the CPU restatement oracle/efe_oracle.py with the same build-defined network.

PARITY UNPINNED: the reference has no runnable semantics for this configuration -- ModelDown rejects the resolution
(/root/reference/src/torchmodel.py:77-82) and check_reward calls the undefined calc_reward_animalai (torchmodel.py:213-214) --
so the oracle here is pinned only by construction (it is the dSprites restatement, which IS pinned, with the layer sizes the
resolution implies and the summed NCHW-broadcast reward).  Tolerances are the fp32 ones of tests/test_gpu_parity.py scaled to
the larger pixel sums (3 x 84 x 84 = 21168 terms per image)."""
import numpy as np
import pytest
import torch

from oracle import philox as PX
from oracle import synth
from oracle import efe_oracle as EO

pytestmark = pytest.mark.gpu
A, C, R = 3, 3, 84
NPIX = C * R * R


def c(t):
    return t.detach().cpu().numpy()


def sumtol(ref):
    """a 21168-term fp32 sum of O(1) values, |sum| up to 2.5e5: a few ulp of the sum plus an absolute floor"""
    return 8e-6 * max(float(np.max(np.abs(ref))), 1.0) + 2e-2


@pytest.fixture(scope='module')
def pair():
    import daimc_amd
    w = synth.make_weights(4321, 1.15, A, C, R)
    m = daimc_amd.ActiveInferenceModel(10, A, 0.0, 1.0, 1.0, colour_channels=C, resolution=R, device='cuda:0', seed=9, init_weights=False)
    m.load_flat_weights(w)
    m.eps_source, m.u_source = PX.normals, PX.uniforms
    orc = EO.OracleModel(w, EO.PhiloxNoise(9), pi_dim=A, channels=C, resolution=R)
    assert not m.parity_pinned
    return m, orc


def test_networks(pair):
    m, orc = pair
    st, M = 5, 5
    s = PX.uniform_fill(3, (M, 10), 50, -1.5, 1.5)
    pi = np.eye(A, dtype=np.float32)[np.arange(M) % A]
    fr = synth.make_frames_rgb(11, M, C, R)
    with torch.no_grad():
        ops1, omean, olv = orc.transition_with_sample(torch.from_numpy(pi), torch.from_numpy(s), PX.PASS_T1, 0, st)
        opo = orc.decoder(torch.from_numpy(s), PX.PASS_D1, 0, st)
        oes, oem, oelv = orc.encoder_with_sample(torch.from_numpy(fr), PX.PASS_E1, 0, st)
        ol, oq, olq = orc.encode_s(torch.from_numpy(s))
    ps1, mean, lv = m.model_mid.transition_with_sample(pi, s, stage=st, pass_=PX.PASS_T1)
    np.testing.assert_allclose(c(mean), omean.numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(ps1), ops1.numpy(), rtol=1e-5, atol=2e-6)
    po = m.model_down.decoder(s, stage=st, pass_=PX.PASS_D1)
    assert po.shape == (M, C, R, R)
    np.testing.assert_allclose(c(po), opo.numpy(), rtol=1e-5, atol=1e-5)
    es, em, elv = m.model_down.encoder_with_sample(fr, stage=st, pass_=PX.PASS_E1)
    np.testing.assert_allclose(c(em), oem.numpy(), rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(c(elv), oelv.numpy(), rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(c(es), oes.numpy(), rtol=1e-5, atol=5e-6)
    logits, q, logq = m.model_top.encode_s(s)
    assert q.shape == (M, A)
    np.testing.assert_allclose(c(q), oq.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c(m.check_reward(opo.numpy())), orc.check_reward(opo).numpy(), rtol=3e-6)


@pytest.mark.parametrize('M,S', [(3, 1), (7, 3)])
def test_calculate_G(pair, M, S):
    m, orc = pair
    st = 11
    s0 = PX.uniform_fill(4, (M, 10), 60 + M, -1.0, 1.0)
    pi0 = np.eye(A, dtype=np.float32)[np.arange(M) % A]
    with torch.no_grad():
        oG, oT, ops1, ops1m, opo1 = orc.calculate_G(torch.from_numpy(s0), torch.from_numpy(pi0), S, st)
    parts = []
    G, T, ps1, ps1m, po1 = m.calculate_G(s0, pi0, samples=S, stage=st, _parts=parts)
    np.testing.assert_allclose(c(ps1), ops1.numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(po1), opo1.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c(T[0]), oT[0].numpy(), atol=sumtol(oT[0].numpy()))
    np.testing.assert_allclose(c(T[1]), oT[1].numpy(), atol=1e-3)
    np.testing.assert_allclose(c(parts[0][0]), orc.last_term2_parts[0].numpy(), atol=sumtol(orc.last_term2_parts[0].numpy()))
    np.testing.assert_allclose(c(G), oG.numpy(), atol=3 * sumtol(oT[0].numpy()))
    Gm, Tm, ps1m2, po1m = m.calculate_G_mean(s0[:A], np.eye(A, dtype=np.float32), stage=st + 1)
    with torch.no_grad():
        oGm = orc.calculate_G_mean(torch.from_numpy(s0[:A]), torch.eye(A), st + 1)[0]
    np.testing.assert_allclose(c(Gm), oGm.numpy(), atol=3 * sumtol(oT[0].numpy()))


def test_rollout_and_posterior(pair):
    """configs[4] shape at reduced size: depth 3, 4 samples, 2 episodes x 3 actions"""
    import daimc_amd
    m, orc = pair
    st, D, S, n = 20, 3, 4, 2
    fr = synth.make_frames_rgb(12, n, C, R)
    o = np.repeat(fr, A, axis=0)
    pi = np.tile(np.eye(A, dtype=np.float32), (n, 1))
    with torch.no_grad():
        oG, oT, opo1 = orc.calculate_G_repeated(torch.from_numpy(o), torch.from_numpy(pi), D, False, S, st)
    G, T, po1 = m.calculate_G_repeated(o, pi, steps=D, samples=S, stage=st)
    np.testing.assert_allclose(c(G), oG.numpy(), atol=D * 3 * sumtol(oT[0].numpy() / D))
    np.testing.assert_allclose(c(po1), opo1.numpy(), rtol=1e-5, atol=5e-5)
    P, logP = m.action_posterior(G, A)
    oP, ologP = EO.softmax_multi_with_log(-oG.numpy(), A)
    np.testing.assert_allclose(c(P), oP, atol=5e-3)
    pi0, logPpi, Ppi, sumG = daimc_amd.plan_actions_batch(m, fr, deepness=D, samples=S, stage=st)
    assert pi0.shape == (n, A) and torch.equal(sumG.reshape(-1), G)
    # sharding invariance holds on the generic path too (global row keys)
    G_hi, _, _ = m.calculate_G_repeated(o[A:], pi[A:], steps=D, samples=S, stage=st, row_offset=A)
    assert torch.equal(G_hi, G[A:])


def test_rollout_at_benchmark_depth_and_samples(pair):
    """BASELINE configs[4] compared at its OWN depth and sample count -- depth 7, 30 MC samples -- on 24 rows (8 episodes x 3 actions;
    the bench runs 96): 7 x 90 x 24 = 15120 decoder images through the same launch groups as the benchmark.  (~20 s of oracle time.)"""
    m, orc = pair
    from conftest import usable_cores
    st, D, S, n = 70, 7, 30, 8
    fr = synth.make_frames_rgb(13, n, C, R)
    o = np.repeat(fr, A, axis=0)
    pi = np.tile(np.eye(A, dtype=np.float32), (n, 1))
    prev = torch.get_num_threads()
    torch.set_num_threads(usable_cores())
    try:
        with torch.no_grad():
            oG, oT, opo1 = orc.calculate_G_repeated(torch.from_numpy(o), torch.from_numpy(pi), D, False, S, st)
    finally:
        torch.set_num_threads(prev)
    G, T, po1 = m.calculate_G_repeated(o, pi, steps=D, samples=S, stage=st)
    per_stage = sumtol(oT[0].numpy() / D)
    np.testing.assert_allclose(c(T[0]), oT[0].numpy(), atol=D * per_stage)
    np.testing.assert_allclose(c(T[1]), oT[1].numpy(), atol=D * 1e-3)
    np.testing.assert_allclose(c(G), oG.numpy(), atol=D * 3 * per_stage)
    np.testing.assert_allclose(c(po1), opo1.numpy(), rtol=1e-5, atol=5e-5)


def test_planner_three_actions(pair):
    """single-episode and lock-step planners with pi_dim 3 (the {1,2} opposite pair of mcts.py:119-124) vs the oracle planner"""
    import daimc_amd
    from oracle import mcts_oracle as MO
    m, orc = pair
    frames = synth.make_frames_rgb(13, 2, C, R)
    p = daimc_amd.MCTS_Params()
    p.repeats, p.simulation_depth, p.use_means, p.threshold, p.samples = 4, 3, False, 0.9, 2
    op = MO.Params(repeats=4, simulation_depth=3, use_means=False, threshold=0.9, samples=2)
    m._stage = 40
    out, visits = daimc_amd.active_inference_mcts_batch(m, torch.from_numpy(frames), p, o_shape=(C, R, R))
    for e in range(2):
        path, reps, explored, all_paths, all_G, root_N = MO.plan(orc, torch.from_numpy(frames[e]), op, 40, episode=e)
        assert out[e][0] == path and out[e][1] == reps and out[e][2] == explored and out[e][3] == all_paths
        np.testing.assert_allclose(np.array(out[e][4]), np.array(all_G), atol=3 * sumtol(np.array(all_G)))
        np.testing.assert_array_equal(visits[e].numpy(), (root_N / root_N.sum()).numpy())
    m._stage = 40
    path1, reps1, explored1, ap1, ag1 = daimc_amd.active_inference_mcts(m, torch.from_numpy(frames[0]), p, o_shape=(C, R, R))
    assert path1 == out[0][0] and [[int(a) for a in q] for q in ap1] == out[0][3]


def test_configs4_full_shape_properties(pair):
    """BASELINE configs[4] per-GPU shape (32 episodes x 3 actions, 30 MC samples, depth 7): determinism, finiteness, term identity"""
    m, _ = pair
    fr = synth.make_frames_rgb(14, 32, C, R)
    o = np.repeat(fr, A, axis=0)
    pi = np.tile(np.eye(A, dtype=np.float32), (32, 1))
    src = m.eps_source
    m.eps_source = None                       # device noise: the production mode
    try:
        G1, t1, _ = m.calculate_G_repeated(o, pi, steps=7, samples=30, stage=0)
        G2, _, _ = m.calculate_G_repeated(o, pi, steps=7, samples=30, stage=0)
    finally:
        m.eps_source = src
    assert torch.equal(G1, G2) and torch.isfinite(G1).all()
    np.testing.assert_allclose(c(-t1[0] + t1[1] + t1[2]), c(G1), rtol=1e-5)


def test_resolution32_networks_vs_reference_fixture():
    """the engine's generic path on the reference's OWN Animal-AI-branch geometry (pi 3, 3 x 32 x 32, decoder 16 -> 16 -> 32 -> 32 with a
    stride-1 third layer, torchmodel.py:77-80) against fixtures captured from the reference's networks: transition, decoder, encoder,
    habit are PINNED at network level for this branch (its EFE terms are not: the reference's reward function does not exist)"""
    import daimc_amd
    from conftest import load_golden
    g = load_golden('nets32_g115')
    A32, C32, R32, seed, st = int(g['pi_dim']), int(g['channels']), int(g['resolution']), int(g['nseed']), int(g['stage'])
    m = daimc_amd.ActiveInferenceModel(10, A32, 0.0, 1.0, 1.0, colour_channels=C32, resolution=R32, device='cuda:0', seed=seed, init_weights=False)
    m.load_flat_weights(synth.make_weights(int(g['wseed']), float(g['gain']), A32, C32, R32))
    M = len(g['s'])
    ps1, mean, lv = m.model_mid.transition_with_sample(g['pi'], g['s'], stage=st, pass_=PX.PASS_T1, eps=PX.normals(seed, M, 10, PX.PASS_T1, 0, st))
    np.testing.assert_allclose(c(mean), g['t_mean'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(lv), g['t_lv'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(ps1), g['t_ps1'], rtol=1e-5, atol=2e-6)
    po = m.model_down.decoder(g['s'], stage=st, pass_=PX.PASS_D1)
    assert po.shape == (M, C32, R32, R32)
    np.testing.assert_allclose(c(po), g['d_po'], rtol=1e-5, atol=1e-5)
    s, emean, elv = m.model_down.encoder_with_sample(g['frames'], stage=st, pass_=PX.PASS_E1, eps=PX.normals(seed, M, 10, PX.PASS_E1, 0, st))
    np.testing.assert_allclose(c(emean), g['e_mean'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(elv), g['e_lv'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(s), g['e_s'], rtol=1e-5, atol=2e-6)
    logits, q, logq = m.model_top.encode_s(g['s'])
    np.testing.assert_allclose(c(logits), g['h_logits'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c(q), g['h_q'], rtol=1e-5, atol=1e-6)
    # the shipped 256-input first encoder Linear of this variant is rejected with the same explanation as the dSprites one
    sd = m.model_down.state_dict(); sd['qs_net.9.weight'] = torch.zeros(256, 256)
    with pytest.raises(ValueError, match='torchmodel.py:94'):
        m.model_down.load_state_dict(sd)
    # a whole EFE call runs on this geometry too (build-defined reward): finite and deterministic
    G1 = m.calculate_G(g['s'][:3], np.eye(3, dtype=np.float32), samples=2, stage=1)[0]
    G2 = m.calculate_G(g['s'][:3], np.eye(3, dtype=np.float32), samples=2, stage=1)[0]
    assert torch.equal(G1, G2) and torch.isfinite(G1).all()


@pytest.mark.parametrize('A2,C2,R2', [(4, 1, 48), (3, 3, 64), (5, 2, 128), (2, 3, 36)])
def test_other_geometries_networks(A2, C2, R2):
    """every strip shape of the LDS-tiled ConvT kernels (two or three rows per strip, short last strips, Win not a divisor of
    the tile) and the final layer's 3 / 2 rows per iteration, against the oracle restatement of the same geometry"""
    import daimc_amd
    w = synth.make_weights(77 + R2, 1.15, A2, C2, R2)
    m = daimc_amd.ActiveInferenceModel(10, A2, 0.0, 1.0, 1.0, colour_channels=C2, resolution=R2, device='cuda:0', seed=5, init_weights=False)
    m.load_flat_weights(w)
    m.eps_source, m.u_source = PX.normals, PX.uniforms
    orc = EO.OracleModel(w, EO.PhiloxNoise(5), pi_dim=A2, channels=C2, resolution=R2)
    st, M = 2, 3
    s = PX.uniform_fill(4, (M, 10), 60, -1.5, 1.5)
    fr = synth.make_frames_rgb(12, M, C2, R2)
    with torch.no_grad():
        opo = orc.decoder(torch.from_numpy(s), PX.PASS_D1, 0, st)
        oes, oem, oelv = orc.encoder_with_sample(torch.from_numpy(fr), PX.PASS_E1, 0, st)
    po = m.model_down.decoder(s, stage=st, pass_=PX.PASS_D1)
    assert po.shape == (M, C2, R2, R2)
    np.testing.assert_allclose(c(po), opo.numpy(), rtol=1e-5, atol=1e-5)
    es, em, elv = m.model_down.encoder_with_sample(fr, stage=st, pass_=PX.PASS_E1)
    np.testing.assert_allclose(c(em), oem.numpy(), rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(c(elv), oelv.numpy(), rtol=1e-5, atol=5e-6)
    # the per-image entropy sum of the fused final layer against the oracle's calculate_G term (one sample, mean mode off)
    pi = np.eye(A2, dtype=np.float32)[np.arange(M) % A2]
    with torch.no_grad():
        oG, oterms = orc.calculate_G(torch.from_numpy(s), torch.from_numpy(pi), 2, st)[:2]
    G, terms = m.calculate_G(s, pi, samples=2, stage=st)[:2]
    np.testing.assert_allclose(c(terms[0]), oterms[0].numpy(), atol=sumtol(oterms[0].numpy()))
    np.testing.assert_allclose(c(terms[1]), oterms[1].numpy(), atol=1e-3)
    np.testing.assert_allclose(c(G), oG.numpy(), atol=3 * sumtol(oterms[0].numpy()))


def test_row_mask_generic_geometry(pair):
    """efe_rows.mask on the generic path (ConvT / final-layer kernels skip dead images): live rows bit-identical"""
    m, _ = pair
    Eps = 4
    M = A * Eps
    s0 = PX.uniform_fill(4, (M, 10), 160, -1.0, 1.0)
    pi0 = np.eye(A, dtype=np.float32)[np.arange(M) % A]
    ref = m.calculate_G(s0, pi0, samples=2, stage=6)
    alive = torch.tensor([0, 1, 1, 0], dtype=torch.uint8, device=m.device)
    from daimc_amd.model import Rows
    out = m.calculate_G(s0, pi0, samples=2, stage=6, rows=Rows(mask=alive, rows_per_entry=A))
    rows = alive.bool().repeat_interleave(A)
    assert torch.equal(out[0][rows], ref[0][rows]) and torch.equal(out[2][rows], ref[2][rows]) and torch.equal(out[4][rows], ref[4][rows])
    assert torch.equal(m.calculate_G(s0, pi0, samples=2, stage=6)[0], ref[0])
    # a compacted call (entries 1, 2 only) -- the generic kernels read the mask at the entry id
    keep = torch.nonzero(alive).flatten()
    kr = (keep[:, None] * A + torch.arange(A, device=m.device)[None]).reshape(-1)
    s0d, pid = torch.from_numpy(s0).to(m.device), torch.from_numpy(pi0).to(m.device)
    cmp_ = m.calculate_G(s0d[kr], pid[kr], samples=2, stage=6, rows=Rows(ids=keep.to(torch.int32), rows_per_entry=A, ids_host=keep.tolist()))
    assert torch.equal(cmp_[0], ref[0][kr]) and torch.equal(cmp_[2], ref[2][kr]) and torch.equal(cmp_[4], ref[4][kr])
    alive2 = alive.clone(); alive2[2] = 0
    cm = m.calculate_G(s0d[kr], pid[kr], samples=2, stage=6, rows=Rows(mask=alive2, ids=keep.to(torch.int32), rows_per_entry=A, ids_host=keep.tolist()))
    assert torch.equal(cm[0][:A], ref[0][kr][:A])


def test_generic_chunking_invariance(pair):
    """launch groups of the generic decoder / encoder (options dec_chunk_g, enc_chunk) do not change a bit of the result"""
    m, _ = pair
    M = 7
    s0 = PX.uniform_fill(4, (M, 10), 170, -1.0, 1.0)
    pi0 = np.eye(A, dtype=np.float32)[np.arange(M) % A]
    ref = m.calculate_G(s0, pi0, samples=3, stage=8)
    try:
        m.set_option('dec_chunk_g', 5)
        m.set_option('enc_chunk', 4)
        out = m.calculate_G(s0, pi0, samples=3, stage=8)
    finally:
        m.set_option('dec_chunk_g', 16384)
        m.set_option('enc_chunk', 32768)
    assert torch.equal(out[0], ref[0]) and torch.equal(out[4], ref[4])
    for k in range(3):
        assert torch.equal(out[1][k], ref[1][k])


@pytest.mark.parametrize('A2,C2,R2', [(3, 3, 84), (4, 1, 48), (3, 3, 64), (5, 2, 128), (2, 3, 36)])
def test_fused_last_layers_equal_separate_launches(A2, C2, R2):
    """k_dec_bg (ConvT3 + final layer + sigmoid + sums in one kernel, option fuse_final_g = 1, the default) against the same layers
    as separate launches (k_convt_p + k_final_g through y3 in HBM): identical images up to the summation order of the 3 x 3 taps, equal
    per-image sums within the fp32 bound -- on every strip shape (Win = 42, 24, 32, 64, 18), including a reward-target row split"""
    import daimc_amd
    w = synth.make_weights(91 + R2, 1.15, A2, C2, R2)
    m = daimc_amd.ActiveInferenceModel(10, A2, 0.0, 1.0, 1.0, colour_channels=C2, resolution=R2, device='cuda:0', seed=6, init_weights=False)
    m.load_flat_weights(w)
    m.eps_source, m.u_source = PX.normals, PX.uniforms
    M, st = 5, 4
    s0 = PX.uniform_fill(4, (M, 10), 61, -1.2, 1.2)
    pi0 = np.eye(A2, dtype=np.float32)[np.arange(M) % A2]
    parts_f, parts_u = [], []
    Gf, Tf, _, _, pof = m.calculate_G(s0, pi0, samples=2, stage=st, _parts=parts_f)
    m.set_option('fuse_final_g', 0)
    try:
        Gu, Tu, _, _, pou = m.calculate_G(s0, pi0, samples=2, stage=st, _parts=parts_u)
    finally:
        m.set_option('fuse_final_g', 1)
    np.testing.assert_allclose(c(pof), c(pou), rtol=0, atol=4e-6)
    tol = sumtol(c(Tu[0]))
    np.testing.assert_allclose(c(Tf[0]), c(Tu[0]), atol=tol)
    np.testing.assert_allclose(c(parts_f[0][0]), c(parts_u[0][0]), atol=tol)
    np.testing.assert_allclose(c(Gf), c(Gu), atol=3 * tol)
    np.testing.assert_allclose(c(Tf[1]), c(Tu[1]), atol=1e-4)       # term1 reads the encoder's view of the stored images (a few ulp apart)


@pytest.mark.parametrize('A2,C2,R2', [(3, 3, 84), (4, 1, 48), (3, 3, 64), (5, 2, 128), (2, 3, 36), (3, 3, 32)])
def test_fused_first_transposed_layers_are_bit_identical(A2, C2, R2):
    """k_convt_12 (ConvT1 + ReLU + ConvT2 + ReLU in one kernel, layer 1's output kept in an LDS ring: option ct_fuse12 = 1, the default)
    against one launch per layer (k_convt_p<1>, k_convt_p<2> through y1 in HBM): the same MFMA sequence per element -- identical bits of
    every output of calculate_G, stored images included -- on every strip shape (base 21, 12, 16, 32, 9, 8: strips of 3, 5, 4, 2, 7, 8
    rows), with a row mask"""
    import daimc_amd
    w = synth.make_weights(95 + R2, 1.15, A2, C2, R2)
    m = daimc_amd.ActiveInferenceModel(10, A2, 0.0, 1.0, 1.0, colour_channels=C2, resolution=R2, device='cuda:0', seed=6, init_weights=False)
    m.load_flat_weights(w)
    m.eps_source, m.u_source = PX.normals, PX.uniforms
    M, st = 7, 4
    s0 = PX.uniform_fill(4, (M, 10), 63, -1.2, 1.2)
    pi0 = np.eye(A2, dtype=np.float32)[np.arange(M) % A2]
    from daimc_amd.model import Rows
    mask = torch.tensor([1, 1, 0, 1, 1, 1, 0], dtype=torch.uint8, device=m.device)
    outs = {}
    try:
        for mode in (1, 0):
            m.set_option('ct_fuse12', mode)
            full = m.calculate_G(s0, pi0, samples=2, stage=st)
            part = m.calculate_G(s0, pi0, samples=2, stage=st, rows=Rows(mask=mask))
            outs[mode] = [full[0], *full[1], full[4], part[0]]
    finally:
        m.set_option('ct_fuse12', 1)
    live = mask.bool()
    for x, y in zip(outs[1][:-1], outs[0][:-1]):
        assert torch.equal(x, y)
    assert torch.equal(outs[1][-1][live], outs[0][-1][live]) and torch.equal(outs[1][-1][live], outs[1][0][live])


def test_reward_upstream_intent_generic(pair):
    """the upstream-intent reward target (option reward_upstream_intent) on the generic geometry: rows 0..2, left half = 1,
    summed over channels -- against oracle.efe_oracle.check_reward_upstream_intent, through check_reward and through calculate_G"""
    m, orc = pair
    st, M = 31, 4
    s0 = PX.uniform_fill(4, (M, 10), 63, -1.0, 1.0)
    pi0 = np.eye(A, dtype=np.float32)[np.arange(M) % A]
    ref = m.calculate_G(s0, pi0, samples=2, stage=st)
    m.set_option('reward_upstream_intent', 1)
    orc.reward_upstream_intent = True
    try:
        with torch.no_grad():
            oG, oT, _, _, opo1 = orc.calculate_G(torch.from_numpy(s0), torch.from_numpy(pi0), 2, st)
            ocr = orc.check_reward(opo1)
        G, T, _, _, po1 = m.calculate_G(s0, pi0, samples=2, stage=st)
        np.testing.assert_allclose(c(T[0]), oT[0].numpy(), atol=sumtol(oT[0].numpy()))
        np.testing.assert_allclose(c(m.check_reward(opo1.numpy())), ocr.numpy(), rtol=3e-6, atol=1e-4)
        assert torch.equal(po1, ref[4])
    finally:
        m.set_option('reward_upstream_intent', 0)
        orc.reward_upstream_intent = False
    assert torch.equal(m.calculate_G(s0, pi0, samples=2, stage=st)[0], ref[0])


@pytest.mark.parametrize('A2,C2,R2', [(3, 3, 84), (4, 1, 48), (3, 3, 64), (5, 2, 128), (2, 3, 36), (3, 3, 32)])
def test_tiled_encoder_layers_equal_direct_kernel(A2, C2, R2):
    """k_conv_e12 (encoder layers 1 + 2 in one kernel, conv1 kept in LDS: option enc_tiled = 2, the default) and k_conv_e (LDS-tiled, one
    launch per layer: 1) against k_conv_g on every layer (0): the same fp32 contraction per output element in the same tap / channel order
    -- bit-identical -- on every strip shape, 1..3 input channels, with a row mask, and against the oracle (via the other tests of this
    file, which run the default)"""
    import daimc_amd
    w = synth.make_weights(93 + R2, 1.15, A2, C2, R2)
    m = daimc_amd.ActiveInferenceModel(10, A2, 0.0, 1.0, 1.0, colour_channels=C2, resolution=R2, device='cuda:0', seed=8, init_weights=False)
    m.load_flat_weights(w)
    M = 7
    fr = synth.make_frames_rgb(16, M, C2, R2)
    outs = {}
    try:
        for mode in (2, 1, 0):          # 2 = the default: layers 1 + 2 in one kernel (k_conv_e12), 1 = k_conv_e per layer, 0 = k_conv_g
            m.set_option('enc_tiled', mode)
            outs[mode] = m.model_down.encoder_with_sample(fr, stage=3, pass_=PX.PASS_E1)
    finally:
        m.set_option('enc_tiled', 2)
    for mode in (1, 0):
        assert all(torch.equal(x, y) for x, y in zip(outs[2], outs[mode])), mode


@pytest.mark.parametrize('fuse', [1, 0])
def test_reserve_no_growth_generic(fuse):
    """efe_rollout_scratch_bytes mirrors the generic path's allocations (with and without the fused last two decoder layers): after
    efe_reserve, calls at that size never grow the arena and stay under the reported size"""
    import daimc_amd
    m = daimc_amd.ActiveInferenceModel(10, A, 0.0, 1.0, 1.0, colour_channels=C, resolution=R, device='cuda:0', seed=2)
    m.set_option('fuse_final_g', fuse)
    need = m.reserve(9, 2, 3)
    st0 = m.arena_stats()
    assert st0['capacity_bytes'] >= need
    o = synth.make_frames_rgb(35, 9, C, R)
    pi = np.eye(A, dtype=np.float32)[np.arange(9) % A]
    for k in range(2):
        m.calculate_G_repeated(o, pi, steps=2, samples=3, stage=10 * k)
    torch.cuda.synchronize()
    st1 = m.arena_stats()
    assert st1['grow_count'] == st0['grow_count'] and st1['capacity_bytes'] == st0['capacity_bytes']
    assert 0 < st1['high_water_bytes'] <= need
