"""Caller-side helpers of the hot path (mirror of /root/reference/src/util.py:46-53)."""
import numpy as np


def softmax_multi_with_log(x, single_values=4, eps=1e-20, temperature=10.0):
    """Action posterior from negated EFE (util.py:46-53,68).  `SM` divides by the temperature, `logSM`
    does not -- kept as the reference computes it.  Host/numpy version for numpy callers; the device
    version is ActiveInferenceModel.action_posterior()."""
    x = np.asarray(x).reshape(-1, single_values)
    x = x - x.max(axis=1, keepdims=True)
    e_x = np.exp(x / temperature)
    tot = e_x.sum(axis=1, keepdims=True)
    return e_x / tot, x - np.log(tot + eps)


def plan_actions_batch(model, frames, deepness=10, samples=5, calc_mean=False, temperature=10.0, generator=None, **kw):
    """The model side of make_batch_dsprites_active_inference (/root/reference/src/util.py:55-70), environment excluded:
    every frame is repeated once per action (row 4i + a), all rows are rolled out `deepness` steps with `samples` MC
    samples in ONE engine call, the summed EFE becomes the action posterior (temperature-10 softmax of -G) and an action
    is sampled per frame.  frames: [n, 1, 64, 64] (or [n, 64, 64, 1] HWC as the environment emits, C = 1).
    -> (pi0 one-hot [n,4], log_Ppi [n,4], Ppi [n,4], sum_G [n,4]) as device tensors."""
    import torch
    f = torch.as_tensor(frames)
    n = f.shape[0]
    A = model.pi_dim
    f = f.reshape(n, model.colour_channels, model.resolution, model.resolution)
    o0_repeated = f.repeat_interleave(A, dim=0)                       # util.py:56-57 (intended row order A*i + a)
    pi_repeated = torch.eye(A, device=model.device).repeat(n, 1)     # util.py:59-60
    sum_G, sum_terms, _ = model.calculate_G_repeated(o0_repeated, pi_repeated, steps=deepness, samples=samples,
                                                     calc_mean=calc_mean, **kw)
    Ppi, log_Ppi = model.action_posterior(sum_G, A, temperature)     # util.py:68
    choices = torch.multinomial(Ppi, 1, generator=generator).squeeze(1)   # util.py:70 (np.random.choice per game)
    pi0 = torch.zeros(n, A, device=Ppi.device)
    pi0[torch.arange(n, device=Ppi.device), choices] = 1.0            # util.py:73-74
    return pi0, log_Ppi, Ppi, sum_G.reshape(n, A)


def make_batch_dsprites_active_inference(games, model, deepness=10, samples=5, calc_mean=False, repeats=1, generator=None):
    """/root/reference/src/util.py:55-80 with the batched device environment (env.Game): observe all games, plan every game's
    action by EFE rollouts in one engine call, act, observe again.  -> (o0, o1, pi0, log_Ppi) as device tensors
    (o0/o1 in the reference's [games,64,64,1] layout)."""
    o0 = games.current_frame_all()
    pi0, log_Ppi, Ppi, _ = plan_actions_batch(model, o0, deepness=deepness, samples=samples, calc_mean=calc_mean, generator=generator)
    games.pi_to_action_all(pi0.argmax(dim=1), repeats=repeats)
    o1 = games.current_frame_all()
    return o0, o1, pi0, log_Ppi
