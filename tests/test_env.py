"""Dynamic-dSprites environment: CPU oracle vs the fixtures captured from the reference `Game`, and
(-m gpu) the HIP kernels vs the same fixtures, bit-exact (integer / exactly-representable state arithmetic)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import env_oracle as EV


def test_env_oracle_vs_reference_fixture():
    g = load_golden('env')
    seed = int(g['seed'])
    assert np.array_equal(EV.new_image_all(seed, np.zeros((6, 7), np.float32), int(g['init_stage'])), g['s_init'])   # constructor state
    assert not g['r_init'].any() and not g['s_init'][:, 6].any()
    s, r = EV.reset(seed, 6, 0)
    assert np.array_equal(s, g['s_reset']) and np.array_equal(r, g['r_reset'])
    bank = EV.sprite_bank()
    s, r = g['s_in'].copy(), g['r_in'].copy()
    assert np.array_equal(EV.render(s, r, bank)[..., 0], g['frames_in'])
    for t in range(len(g['actions'])):
        ch = EV.step(seed, s, r, g['actions'][t], int(g['repeats']), 1 + t)
        assert np.array_equal(ch, g['changed'][t])
        assert np.array_equal(s, g['states'][t]) and np.array_equal(r, g['last_r'][t])
        assert np.array_equal(EV.render(s, r, bank)[..., 0], g['frames'][t])
    assert g['changed'].sum() >= 4          # the fixture exercises finished rounds (latent resampling + reward restart)


def test_env_render_rejects_out_of_range_reward():
    s, r = EV.reset(1, 2, 0)
    r[1] = 1.5
    with pytest.raises(ValueError):
        EV.render(s, r, EV.sprite_bank())


@pytest.mark.gpu
def test_env_kernels_vs_reference_fixture():
    import daimc_amd
    g = load_golden('env')
    games = daimc_amd.Game(6, device='cuda:0', seed=int(g['seed']), init_stage=int(g['init_stage']))
    # the constructor is new_image_all (game_environment.py:21): fresh latents, reward 0, last_r 0
    assert np.array_equal(games.current_s.cpu().numpy(), g['s_init']) and np.array_equal(games.last_r.cpu().numpy(), g['r_init'])
    games.randomize_environment_all(stage=0)
    assert np.array_equal(games.current_s.cpu().numpy(), g['s_reset']) and np.array_equal(games.last_r.cpu().numpy(), g['r_reset'])
    games.current_s.copy_(torch.from_numpy(g['s_in'])); games.last_r.copy_(torch.from_numpy(g['r_in']))
    f = games.current_frame_all()
    assert f.shape == (6, 64, 64, 1)
    assert np.array_equal(f.cpu().numpy()[..., 0], g['frames_in'])
    for t in range(len(g['actions'])):
        ch = games.pi_to_action_all(g['actions'][t], repeats=int(g['repeats']), stage=1 + t)
        assert np.array_equal(ch.cpu().numpy(), g['changed'][t])
        assert np.array_equal(games.current_s.cpu().numpy(), g['states'][t])
        assert np.array_equal(games.last_r.cpu().numpy(), g['last_r'][t])
        assert np.array_equal(games.current_frame_all().cpu().numpy()[..., 0], g['frames'][t])
    games.last_r[2] = -1.5
    with pytest.raises(ValueError):
        games.current_frame_all()
    with pytest.raises(ValueError):
        games.pi_to_action_all([0, 1, 2, 3, 4, 0])


@pytest.mark.gpu
def test_closed_loop_batch_producer():
    """make_batch_dsprites_active_inference (util.py:55-80) end to end on the device: observe -> EFE rollouts -> act -> observe"""
    import daimc_amd
    m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=5)
    games = daimc_amd.Game(16, model=m, seed=9)
    s_before = games.current_s.clone()
    o0, o1, pi0, logP = daimc_amd.make_batch_dsprites_active_inference(games, m, deepness=2, samples=2, repeats=3)
    assert o0.shape == (16, 64, 64, 1) and o1.shape == (16, 64, 64, 1) and pi0.shape == (16, 4) and logP.shape == (16, 4)
    assert torch.all(pi0.sum(1) == 1) and torch.isfinite(logP).all()
    # every game either moved, was already at a wall for its action, or finished a round; reward bar decays by 0.95^3 otherwise
    moved = (games.current_s != s_before).any(dim=1)
    a = pi0.argmax(1)
    wall = ((a == 1) & (s_before[:, 5] == 0)) | ((a == 2) & (s_before[:, 4] == 31)) | ((a == 3) & (s_before[:, 4] == 0))
    assert bool((moved | wall).all())
    # sharded games reproduce the unsharded ones (global game keys)
    g_all = daimc_amd.Game(8, model=m, seed=3); g_all.randomize_environment_all(stage=4)
    g_hi = daimc_amd.Game(4, model=m, seed=3, game_offset=4); g_hi.randomize_environment_all(stage=4)
    assert torch.equal(g_all.current_s[4:], g_hi.current_s) and torch.equal(g_all.last_r[4:], g_hi.last_r)
