"""Batched Dynamic-dSprites environment on the device: the `Game` of
/root/reference/src/game_environment.py for many games at once -- state, stepping and frame rendering are HIP
kernels (csrc/kernels.hip: k_env_reset / k_env_step / k_env_render) behind `efe_env_*` of the C ABI, so a planner
can close the observe -> plan -> act loop for thousands of parallel episodes without leaving the GPU.

The reference's own quirks are kept where results depend on them (the image index is dot(s[:6], [1,3,6,40,32,32]),
game_environment.py:25,39-42); the per-index Python API is replaced by `*_all` calls.  The dSprites archive is not
shipped with the reference: pass `imgs` (uint8 [N,64,64]) or use the synthetic bank."""
import ctypes as C

import torch

from . import _lib
from .model import _Engine, _ptr


def synthetic_sprite_bank(n=3581):
    """deterministic binary sprites, one filled box per index (same rule as oracle/env_oracle.py::sprite_bank)"""
    imgs = torch.zeros(n, 64, 64, dtype=torch.uint8)
    for i in range(n):
        side = 6 + (i * 7) % 18
        y = 3 + (i * 13) % (61 - side)
        x = (i * 29) % (64 - side)
        imgs[i, y:y + side, x:x + side] = 1
    return imgs


class Game:
    def __init__(self, games_no, *, model=None, device=None, imgs=None, seed=0, game_offset=0, init_stage=None):
        self.games_no = int(games_no)
        if model is not None:
            self._engine = model._engine
        else:
            idx = torch.device(device).index or 0 if device is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
            self._engine = _Engine(idx)
        self.device = self._engine.device
        bank = synthetic_sprite_bank() if imgs is None else torch.as_tensor(imgs)
        self.imgs = bank.reshape(-1, 64, 64).to(device=self.device, dtype=torch.uint8).contiguous()
        self.s_sizes = torch.tensor([1, 3, 6, 40, 32, 32])
        self.s_bases = torch.tensor([1, 3, 6, 40, 32, 32])
        self.s_dim = 7
        self.seed, self.game_offset, self._stage = int(seed), int(game_offset), 0
        self.current_s = torch.zeros(self.games_no, 7, device=self.device)
        self.last_r = torch.zeros(self.games_no, device=self.device)
        self.new_image_all(init_stage)  # game_environment.py:21: fresh latents, reward 0, last_r 0 (randomisation is an explicit call, train.py:107)

    def _nz(self, stage):
        if stage is None:
            stage = self._stage
            self._stage += 1
        return _lib.EfeNoise(self.seed, int(stage), 9, 0, self.game_offset)

    def new_image_all(self, stage=None):
        """game_environment.py:83-88: new latents for every game; accumulated reward and last_r are kept"""
        e = self._engine
        nz = self._nz(stage)
        e.check(e.lib.efe_env_new_image(e.ctx, _ptr(self.current_s), self.games_no, C.byref(nz), e.stream()))

    def randomize_environment_all(self, stage=None):
        """game_environment.py:72-75"""
        e = self._engine
        nz = self._nz(stage)
        e.check(e.lib.efe_env_reset(e.ctx, _ptr(self.current_s), _ptr(self.last_r), self.games_no, C.byref(nz), e.stream()))

    def pi_to_action_all(self, actions, repeats=1, stage=None):
        """pi_to_action(actions[i], i, repeats) for every game (game_environment.py:154-169) -> round_changed [games] bool"""
        e = self._engine
        a = torch.as_tensor(actions).to(device=self.device, dtype=torch.int32).contiguous()
        if a.numel() != self.games_no or bool(((a < 0) | (a > 3)).any()):
            raise ValueError('Invalid action')
        changed = torch.empty(self.games_no, dtype=torch.int32, device=self.device)
        nz = self._nz(stage)
        e.check(e.lib.efe_env_step(e.ctx, _ptr(self.current_s), _ptr(self.last_r), C.c_void_p(a.data_ptr()), self.games_no, int(repeats),
                                   C.byref(nz), C.c_void_p(changed.data_ptr()), e.stream()))
        return changed.bool()

    def current_frame_nchw(self):
        """[games,1,64,64] frames for the model (the reshape of the HWC frame is free for one channel, mcts.py:158)"""
        e = self._engine
        frames = torch.empty(self.games_no, 1, 64, 64, device=self.device)
        err = torch.empty(self.games_no, dtype=torch.int32, device=self.device)
        e.check(e.lib.efe_env_render(e.ctx, _ptr(self.current_s), _ptr(self.last_r), C.c_void_p(self.imgs.data_ptr()), self.imgs.shape[0],
                                     _ptr(frames), C.c_void_p(err.data_ptr()), self.games_no, e.stream()))
        if bool(err.any()):
            raise ValueError(f'Error: Reward: {self.last_r[err.bool()][0].item()}')      # game_environment.py:53
        return frames

    def current_frame_all(self):
        """[games,64,64,1] like the reference (game_environment.py:62-66)"""
        return self.current_frame_nchw().reshape(self.games_no, 64, 64, 1)

    def tick_all(self):
        self.last_r *= 0.95

    def get_reward_all(self):
        return self.current_s[:, 6]

    def find_move_all(self, randomness):
        """game_environment.py:94-104: the hand-coded good policy, [games,4]"""
        right = 0.5 * (1.0 - randomness / 2.0)
        wrong = 0.5 * randomness / 2.0
        sq = torch.tensor([right, wrong, wrong, right], device=self.device)
        el = torch.tensor([right, wrong, right, wrong], device=self.device)
        return torch.where((self.current_s[:, 1] < 0.5)[:, None], sq[None, :], el[None, :])
