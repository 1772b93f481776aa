"""ORACLE -- test infrastructure only.  Never imported by the product path
(`deep-active-inference-mc_amd/`); only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may use it.

CPU restatement (PyTorch-CPU functional ops, fp32, same op order) of the
reference's expected-free-energy hot path, `/root/reference/src/torchmodel.py`
and `/root/reference/src/torchutils.py`.  Every function cites the reference
lines it follows.  Differences from the shipped port are exactly the three
sanctioned shim items of SURVEY.md section 0 / appendix C:
  * first encoder Linear has 576 inputs (64*3*3), not 256 (`torchmodel.py:94`),
  * `self.precision` = float32 (`torchmodel.py:355-359`),
  * noise is *injected*: dropout masks / normals / categorical uniforms come
    from the addressable Philox stream of `oracle/philox.py` instead of torch's
    global generator (or, with `noise="torch"`, from torch's own generator --
    used only when timing the CPU baseline so the op mix equals the reference's).

Pinned against the shimmed reference itself by `oracle/make_golden.py` ->
`tests/golden/*.npz` (see tests/test_oracle_golden.py).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

from . import philox as PX

LOG_2_PI_E = np.log(2.0 * np.pi * np.e)   # torchutils.py:19 (numpy float64 scalar)


# --------------------------------------------------------------------------
# helpers: torchutils.py
# --------------------------------------------------------------------------
def entropy_normal_from_logvar(logvar):
    """torchutils.py:22-23"""
    return 0.5 * (LOG_2_PI_E + logvar)


def entropy_bernoulli(p, displacement=0.00001):
    """torchutils.py:26-27"""
    return -(1 - p) * torch.log(displacement + 1 - p) - p * torch.log(displacement + p)


def log_bernoulli(x, p, displacement=0.00001):
    """torchutils.py:30-31"""
    return x * torch.log(displacement + p) + (1 - x) * torch.log(displacement + 1 - p)


def calc_reward(o, resolution=64):
    """torchutils.py:34-37 (NCHW broadcast semantics replicated)."""
    perfect_reward = torch.zeros((3, resolution, 1), dtype=torch.float32)
    perfect_reward[:, :int(resolution / 2)] = 1.0
    return log_bernoulli(o[:, 0:3, 0:resolution, :], perfect_reward)


def check_reward(o):
    """torchmodel.py:210-212 (resolution 64 branch)."""
    return torch.mean(calc_reward(o), dim=[1, 2, 3]) * 10.0


def check_reward_generic(o):
    """BUILD-DEFINED reward of the non-dSprites geometries (the reference's `calc_reward_animalai`,
    torchmodel.py:213-214, does not exist): the NCHW-broadcast target of torchutils.py:34-37 -- 1 for image rows
    h < H/2, 0 below -- applied to every channel, summed over [1,2,3] like the reference's resolution-32 branch."""
    H = o.shape[2]
    target = torch.zeros((1, 1, H, 1), dtype=torch.float32)
    target[:, :, :H // 2] = 1.0
    return torch.sum(log_bernoulli(o, target), dim=[1, 2, 3])


def check_reward_upstream_intent(o, generic=False):
    """The reward the UPSTREAM code means, kept beside the replicated quirk as an engine option
    (`reward_upstream_intent`): `calc_reward` (torchutils.py:34-37) was written for NHWC observations, where `o[:, 0:3, 0:res, :]` is the
    top three image rows (the reward bar of game_environment.py:44-54) and `perfect_reward[:, :res/2] = 1` marks their LEFT half.
    Restated for this repository's NCHW tensors: log_bernoulli over rows 0..2 with target 1 for columns < W/2, every channel;
    dSprites: mean over those 3*W*C elements * 10 (torchmodel.py:212); generic geometries: their sum (torchmodel.py:214 form)."""
    W = o.shape[3]
    target = torch.zeros((1, 1, 1, W), dtype=torch.float32)
    target[..., :W // 2] = 1.0
    lb = log_bernoulli(o[:, :, 0:3, :], target)
    return torch.sum(lb, dim=[1, 2, 3]) if generic else torch.mean(lb, dim=[1, 2, 3]) * 10.0


def softmax_multi_with_log(x, single_values=4, eps=1e-20, temperature=10.0):
    """util.py:46-53 (numpy; note logSM is NOT log(SM) -- replicated as is)."""
    x = x.reshape(-1, single_values)
    x = x - np.max(x, 1).reshape(-1, 1)
    e_x = np.exp(x / temperature)
    SM = e_x / e_x.sum(axis=1).reshape(-1, 1)
    logSM = x - np.log(e_x.sum(axis=1).reshape(-1, 1) + eps)
    return SM, logSM


# --------------------------------------------------------------------------
# noise providers
# --------------------------------------------------------------------------
class PhiloxNoise:
    """Addressable noise; identical keying to csrc/philox.h."""

    def __init__(self, seed, row_offset=0):
        self.seed = int(seed)
        self.row_offset = int(row_offset)

    def mask(self, tag, rows, n_feat, pas, sample, stage, row_offset=None, fc4_perm=False):
        ro = self.row_offset if row_offset is None else row_offset
        m = PX.dropout_mask(self.seed, tag, rows, n_feat, pas, sample, stage, ro)
        if fc4_perm:
            # the engine keys the last dense layer's mask in its NHWC order f' = p*64 + c;
            # the reference feature index is c*P + p (Unflatten(1,(64,B,B)), torchmodel.py:119), P = B*B = n_feat / 64
            m = m.reshape(rows, n_feat // 64, 64).transpose(0, 2, 1).reshape(rows, n_feat)
        return torch.from_numpy(np.ascontiguousarray(m))

    def eps(self, rows, n, pas, sample, stage, row_offset=None):
        ro = self.row_offset if row_offset is None else row_offset
        return torch.from_numpy(PX.normals(self.seed, rows, n, pas, sample, stage, ro))

    def uniform(self, rows, pas, sample, stage, row_offset=None):
        ro = self.row_offset if row_offset is None else row_offset
        return PX.uniforms(self.seed, rows, pas, sample, stage, ro)


class TorchNoise:
    """torch's own generator (what the reference does); for CPU-baseline timing only."""

    def mask(self, tag, rows, n_feat, pas, sample, stage, row_offset=None, fc4_perm=False):
        return torch.empty(rows, n_feat).bernoulli_(0.5) * 2.0

    def eps(self, rows, n, pas, sample, stage, row_offset=None):
        return torch.randn(rows, n)

    def uniform(self, rows, pas, sample, stage, row_offset=None):
        return torch.rand(rows).numpy()


def categorical_from_uniform(probs, u):
    """Inverse-CDF stand-in for torch.multinomial(probs, 1) (torchmodel.py:364,379) with the
    reference's bare-except fallback: invalid probabilities -> action 0.
    probs: 1-D float32 array, u in (0,1). Returns (action, valid)."""
    p = np.asarray(probs, dtype=np.float32)
    tot = np.float32(0.0)
    bad = False
    for v in p:
        if not np.isfinite(v) or v < 0:
            bad = True
        tot = np.float32(tot + v)
    if bad or not (tot > 0):
        return 0, False
    thr = np.float32(np.float32(u) * tot)
    acc = np.float32(0.0)
    for k, v in enumerate(p):
        acc = np.float32(acc + v)
        if thr < acc:
            return k, True
    return len(p) - 1, True


# --------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------
class OracleModel:
    """Functional restatement of ActiveInferenceModel (torchmodel.py:149-393) for
    s_dim=10, pi_dim=4, 1x64x64 observations.  `weights` maps the reference's
    state_dict keys prefixed by 'top.', 'mid.', 'down.' to float32 arrays.

    `channels` / `resolution` other than (1, 64) select the BUILD-DEFINED geometry of SURVEY 8a-13 (BASELINE configs[4]):
    the same layer list with the sizes the resolution implies and check_reward_generic -- the reference cannot run it
    (torchmodel.py:77-82, 213-214), so for that geometry this restatement is the only oracle: PARITY UNPINNED."""

    def __init__(self, weights, noise, s_dim=10, pi_dim=4, channels=1, resolution=64):
        self.w = {k: torch.as_tensor(np.asarray(v, dtype=np.float32)) for k, v in weights.items()}
        self.noise = noise
        self.s_dim = s_dim
        self.pi_dim = pi_dim
        self.channels, self.resolution = channels, resolution
        self.last_stride = 1 if resolution == 32 else 2          # torchmodel.py:77-80
        self.base = resolution // 2 if resolution == 32 else resolution // 4
        self.generic = (channels, resolution) != (1, 64)
        self.reward_upstream_intent = False         # engine option of the same name: see check_reward_upstream_intent
        self.pi_one_hot = torch.eye(pi_dim)      # torchmodel.py:164-165

    def check_reward(self, o):
        if self.reward_upstream_intent:
            return check_reward_upstream_intent(o, self.generic)
        return check_reward_generic(o) if self.generic else check_reward(o)

    # ---- ModelTop.encode_s (torchmodel.py:27-31); no dropout --------------
    def encode_s(self, s0):
        w = self.w
        h = F.relu(F.linear(s0, w['top.qpi_net.0.weight'], w['top.qpi_net.0.bias']))
        h = F.relu(F.linear(h, w['top.qpi_net.2.weight'], w['top.qpi_net.2.bias']))
        logits_pi = F.linear(h, w['top.qpi_net.4.weight'], w['top.qpi_net.4.bias'])
        q_pi = F.softmax(logits_pi, dim=-1)
        log_q_pi = torch.log(q_pi + 1e-20)
        return logits_pi, q_pi, log_q_pi

    # ---- ModelMid (torchmodel.py:41-66) ------------------------------------
    def transition(self, pi, s0, pas, sample, stage, ro=None):
        w = self.w
        M = s0.shape[0]
        h = torch.cat([pi, s0], dim=1)
        for li, idx in enumerate((0, 3, 6)):
            h = F.relu(F.linear(h, w[f'mid.ps_net.{idx}.weight'], w[f'mid.ps_net.{idx}.bias']))
            h = h * self.noise.mask(PX.TAG_MID + li, M, 512, pas, sample, stage, ro)
        out = F.linear(h, w['mid.ps_net.9.weight'], w['mid.ps_net.9.bias'])
        mean, logvar = torch.split(out, self.s_dim, dim=1)
        return mean, logvar

    def reparameterize(self, mean, logvar, pas, sample, stage, ro=None):
        """torchmodel.py:54-56 / 130-132"""
        eps = self.noise.eps(mean.shape[0], mean.shape[1], pas, sample, stage, ro)
        return eps * torch.exp(logvar * 0.5) + mean

    def transition_with_sample(self, pi, s0, pas, sample, stage, ro=None):
        mean, logvar = self.transition(pi, s0, pas, sample, stage, ro)
        ps1 = self.reparameterize(mean, logvar, pas, sample, stage, ro)
        return ps1, mean, logvar

    # ---- ModelDown.decoder (torchmodel.py:106-128,139-141) -----------------
    def decoder(self, s, pas, sample, stage, ro=None):
        w = self.w
        M = s.shape[0]
        h = s
        for li, idx in enumerate((0, 3, 6, 9)):
            h = F.relu(F.linear(h, w[f'down.po_net.{idx}.weight'], w[f'down.po_net.{idx}.bias']))
            nf = h.shape[1]
            h = h * self.noise.mask(PX.TAG_DEC + li, M, nf, pas, sample, stage, ro, fc4_perm=(li == 3))
        h = h.reshape(M, 64, self.base, self.base)
        h = F.relu(F.conv_transpose2d(h, w['down.po_net.13.weight'], w['down.po_net.13.bias'], stride=1, padding=1))
        h = F.relu(F.conv_transpose2d(h, w['down.po_net.15.weight'], w['down.po_net.15.bias'], stride=2, padding=1, output_padding=1))
        ls = getattr(self, 'last_stride', 2)
        h = F.relu(F.conv_transpose2d(h, w['down.po_net.17.weight'], w['down.po_net.17.bias'], stride=ls, padding=1, output_padding=ls - 1))
        h = torch.sigmoid(F.conv_transpose2d(h, w['down.po_net.19.weight'], w['down.po_net.19.bias'], stride=1, padding=1))
        return h

    # ---- ModelDown.encoder (torchmodel.py:84-104,134-137) ------------------
    def encoder(self, o, pas, sample, stage, ro=None):
        w = self.w
        M = o.shape[0]
        h = o
        for idx in (0, 2, 4, 6):
            h = F.relu(F.conv2d(h, w[f'down.qs_net.{idx}.weight'], w[f'down.qs_net.{idx}.bias'], stride=2))
        h = h.reshape(M, -1)                       # Flatten: c*9 + h*3 + w -> 576
        for li, idx in enumerate((9, 12, 15)):
            h = F.relu(F.linear(h, w[f'down.qs_net.{idx}.weight'], w[f'down.qs_net.{idx}.bias']))
            h = h * self.noise.mask(PX.TAG_ENC + li, M, 256, pas, sample, stage, ro)
        out = F.linear(h, w['down.qs_net.18.weight'], w['down.qs_net.18.bias'])
        mean, logvar = torch.split(out, self.s_dim, dim=1)
        return mean, logvar

    def encoder_with_sample(self, o, pas, sample, stage, ro=None):
        mean, logvar = self.encoder(o, pas, sample, stage, ro)
        s = self.reparameterize(mean, logvar, pas, sample, stage, ro)
        return s, mean, logvar

    # ---- calculate_G (torchmodel.py:270-300) -------------------------------
    def calculate_G(self, s0, pi0, samples, stage, ro=None):
        M = s0.shape[0]
        term0 = torch.zeros(M)
        term1 = torch.zeros(M)
        for i in range(samples):
            ps1, ps1_mean, ps1_logvar = self.transition_with_sample(pi0, s0, PX.PASS_T1, i, stage, ro)
            po1 = self.decoder(ps1, PX.PASS_D1, i, stage, ro)
            qs1, _, qs1_logvar = self.encoder_with_sample(po1, PX.PASS_E1, i, stage, ro)
            logpo1 = self.check_reward(po1)
            term0 += logpo1
            term1 += -torch.sum(entropy_normal_from_logvar(ps1_logvar) + entropy_normal_from_logvar(qs1_logvar), dim=1)
        term0 /= float(samples)
        term1 /= float(samples)

        term2_1 = torch.zeros(M)
        term2_2 = torch.zeros(M)
        for j in range(samples):
            po1_temp1 = self.decoder(self.transition_with_sample(pi0, s0, PX.PASS_T2, j, stage, ro)[0], PX.PASS_D2A, j, stage, ro)
            term2_1 += torch.sum(entropy_bernoulli(po1_temp1), dim=[1, 2, 3])
            # ps1_mean / ps1_logvar leak from the LAST loop-1 iteration (torchmodel.py:291)
            po1_temp2 = self.decoder(self.reparameterize(ps1_mean, ps1_logvar, PX.PASS_D2B, j, stage, ro), PX.PASS_D2B, j, stage, ro)
            term2_2 += torch.sum(entropy_bernoulli(po1_temp2), dim=[1, 2, 3])
        term2_1 /= float(samples)
        term2_2 /= float(samples)
        term2 = term2_1 - term2_2
        G = -term0 + term1 + term2
        self.last_term2_parts = (term2_1, term2_2)
        return G, [term0, term1, term2], ps1, ps1_mean, po1

    # ---- calculate_G_mean (torchmodel.py:302-327) --------------------------
    def calculate_G_mean(self, s0, pi0, stage, ro=None):
        _, ps1_mean, ps1_logvar = self.transition_with_sample(pi0, s0, PX.PASS_T1, 0, stage, ro)
        po1 = self.decoder(ps1_mean, PX.PASS_D1, 0, stage, ro)
        _, qs1_mean, qs1_logvar = self.encoder_with_sample(po1, PX.PASS_E1, 0, stage, ro)
        term0 = self.check_reward(po1)
        term1 = -torch.sum(entropy_normal_from_logvar(ps1_logvar) + entropy_normal_from_logvar(qs1_logvar), dim=1)
        po1_temp1 = self.decoder(self.transition_with_sample(pi0, s0, PX.PASS_T2, 0, stage, ro)[1], PX.PASS_D2A, 0, stage, ro)
        term2_1 = torch.sum(entropy_bernoulli(po1_temp1), dim=[1, 2, 3])
        po1_temp2 = self.decoder(self.reparameterize(ps1_mean, ps1_logvar, PX.PASS_D2B, 0, stage, ro), PX.PASS_D2B, 0, stage, ro)
        term2_2 = torch.sum(entropy_bernoulli(po1_temp2), dim=[1, 2, 3])
        term2 = term2_1 - term2_2
        G = -term0 + term1 + term2
        self.last_term2_parts = (term2_1, term2_2)
        return G, [term0, term1, term2], ps1_mean, po1

    # ---- calculate_G_repeated (torchmodel.py:227-245) ----------------------
    def calculate_G_repeated(self, o, pi, steps, calc_mean, samples, stage0, ro=None):
        qs0_mean, qs0_logvar = self.encoder(o, PX.PASS_ROOT, 0, stage0, ro)
        qs0 = self.reparameterize(qs0_mean, qs0_logvar, PX.PASS_ROOT, 0, stage0, ro)
        M = o.shape[0]
        sum_terms = [torch.zeros(M) for _ in range(3)]
        sum_G = torch.zeros(M)
        s0_temp = qs0_mean if calc_mean else qs0
        po1 = None
        for t in range(steps):
            G, terms, s1, ps1_mean, po1 = self.calculate_G(s0_temp, pi, samples, stage0 + t, ro)
            for i in range(3):
                sum_terms[i] += terms[i]
            sum_G += G
            s0_temp = ps1_mean if calc_mean else s1
        return sum_G, sum_terms, po1

    # ---- calculate_G_4_repeated (torchmodel.py:247-268) --------------------
    def calculate_G_4_repeated(self, o, steps, calc_mean, samples, stage0, ro=None):
        qs0_mean, qs0_logvar = self.encoder(o, PX.PASS_ROOT, 0, stage0, ro)
        qs0 = self.reparameterize(qs0_mean, qs0_logvar, PX.PASS_ROOT, 0, stage0, ro)
        sum_terms = [torch.zeros(4) for _ in range(3)]
        sum_G = torch.zeros(4)
        s0_temp = qs0_mean if calc_mean else qs0
        po1 = None
        for t in range(steps):
            if calc_mean:
                G, terms, ps1_mean, po1 = self.calculate_G_mean(s0_temp, self.pi_one_hot, stage0 + t, ro)
            else:
                G, terms, s1, ps1_mean, po1 = self.calculate_G(s0_temp, self.pi_one_hot, samples, stage0 + t, ro)
            for i in range(3):
                sum_terms[i] += terms[i]
            sum_G += G
            s0_temp = ps1_mean if calc_mean else s1
        return sum_G, sum_terms, po1

    # ---- calculate_G_given_trajectory (torchmodel.py:329-352) --------------
    def calculate_G_given_trajectory(self, s0_traj, ps1_traj, ps1_mean_traj, ps1_logvar_traj, pi0_traj, stage, ro=None):
        po1 = self.decoder(ps1_traj, PX.PASS_D1, 0, stage, ro)
        qs1, _, qs1_logvar = self.encoder_with_sample(po1, PX.PASS_E1, 0, stage, ro)
        term0 = self.check_reward(po1)
        term1 = -torch.sum(entropy_normal_from_logvar(ps1_logvar_traj) + entropy_normal_from_logvar(qs1_logvar), dim=1)
        po1_temp1 = self.decoder(self.transition_with_sample(pi0_traj, s0_traj, PX.PASS_T2, 0, stage, ro)[0], PX.PASS_D2A, 0, stage, ro)
        term2_1 = torch.sum(entropy_bernoulli(po1_temp1), dim=[1, 2, 3])
        po1_temp2 = self.decoder(self.reparameterize(ps1_mean_traj, ps1_logvar_traj, PX.PASS_D2B, 0, stage, ro), PX.PASS_D2B, 0, stage, ro)
        term2_2 = torch.sum(entropy_bernoulli(po1_temp2), dim=[1, 2, 3])
        term2 = term2_1 - term2_2
        return -term0 + term1 + term2

    # ---- mcts_step_simulate (torchmodel.py:354-393) ------------------------
    def mcts_step_simulate(self, starting_s, depth, use_means, stage, episode=0):
        """One episode.  Noise rows: habit/transition steps use global row `episode`
        (sample = t); the trajectory batch uses global rows episode*depth + t."""
        s0 = torch.zeros((depth, self.s_dim))
        ps1 = torch.zeros((depth, self.s_dim))
        ps1_mean = torch.zeros((depth, self.s_dim))
        ps1_logvar = torch.zeros((depth, self.s_dim))
        pi0 = torch.zeros((depth, self.pi_dim))
        s0[0] = starting_s
        Qpi_t_to_return = None
        for t in range(depth):
            q = self.encode_s(s0[t].unsqueeze(0))[1][0]
            u = self.noise.uniform(1, PX.PASS_HABIT, t, stage, episode)[0]
            a, valid = categorical_from_uniform(q.numpy(), u)
            pi0[t, a] = 1.0
            if t == 0:
                Qpi_t_to_return = q if valid else pi0[0].clone()
            n_ps1, n_mean, n_logvar = self.transition_with_sample(pi0[t].unsqueeze(0), s0[t].unsqueeze(0), PX.PASS_SIM, t, stage, episode)
            ps1[t] = n_ps1[0]
            ps1_mean[t] = n_mean[0]
            ps1_logvar[t] = n_logvar[0]
            if t + 1 < depth:
                s0[t + 1] = n_mean[0] if use_means else n_ps1[0]
        Gt = self.calculate_G_given_trajectory(s0, ps1, ps1_mean, ps1_logvar, pi0, stage, episode * depth)
        return torch.mean(Gt).item(), pi0, Qpi_t_to_return
