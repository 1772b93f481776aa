"""ORACLE -- test infrastructure only.  CPU restatement (numpy, per-game loops) of the Dynamic-dSprites environment
of /root/reference/src/game_environment.py, rows the engine mirrors in csrc/kernels.hip (k_env_*):
  new_image_all (:83-88), randomize_environment_all (:72-75), tick (:113-117), up/down/left/right (:119-152), pi_to_action (:154-169),
  new_image (:84-87), s_to_index / s_to_o (:39-54; the port's index rule dot(s, [1,3,6,40,32,32]) is replicated as is).
Random latents come from the Philox stream (tag 0x60, pass 9) instead of torch's global generator.
Pinned against the reference Game itself (patched np.load / torch.randint / torch.rand) by oracle/make_golden.py."""
import numpy as np
from . import philox as PX

TAG_ENV, PASS_ENV = 0x60, 9
SIZES = (1, 3, 6, 40, 32, 32)
BASES = np.array([1, 3, 6, 40, 32, 32], dtype=np.float32)


def env_u(seed, game, stage, blk):
    k0, k1 = PX._key(seed)
    x0, _, _, _ = PX.philox4x32_10(np.uint64(blk) | (np.uint64(TAG_ENV) << np.uint64(16)), np.uint64(game),
                                   np.uint64(PX.stream_id(PASS_ENV, 0)), np.uint64(stage), k0, k1)
    return PX._u01(np.asarray(x0, dtype=np.uint32))


def env_randint(seed, game, stage, latent):
    size = SIZES[latent]
    v = int(np.float32(env_u(seed, game, stage, latent)) * np.float32(size))
    return min(v, size - 1)


def reset(seed, n_games, stage, game_offset=0):
    s = np.zeros((n_games, 7), dtype=np.float32)
    last_r = np.zeros(n_games, dtype=np.float32)
    for e in range(n_games):
        g = game_offset + e
        for k in range(6):
            s[e, k] = env_randint(seed, g, stage, k)
        s[e, 6] = np.float32(-10.0) + np.float32(env_u(seed, g, stage, 6)) * np.float32(20.0)
        last_r[e] = np.float32(-1.0) + np.float32(env_u(seed, g, stage, 7)) * np.float32(2.0)
    return s, last_r


def new_image_all(seed, s, stage, game_offset=0):
    """new_image_all (:83-88), in place: fresh latents for every game; the accumulated reward (slot 6) is carried over.
    What the reference constructor calls (:21) on a zero state: reward 0, last_r 0."""
    for e in range(len(s)):
        for k in range(6):
            s[e, k] = env_randint(seed, game_offset + e, stage, k)
    return s


def step(seed, s, last_r, actions, repeats, stage, game_offset=0):
    """in-place pi_to_action for every game; returns round_changed flags"""
    changed = np.zeros(len(s), dtype=bool)
    for e in range(len(s)):
        pi = int(actions[e])
        for _ in range(repeats):
            last_r[e] = np.float32(last_r[e] * np.float32(0.95))
            if pi == 0:
                s[e, 5] += 1.0
                if s[e, 5] >= 32:
                    x = s[e, 4]
                    if s[e, 1] < 0.5:
                        last_r[e] = (15.0 - x) / 16.0 if x > 15 else (16.0 - x) / 16.0
                    else:
                        last_r[e] = (x - 15.0) / 16.0 if x > 15 else (x - 16.0) / 16.0
                    # new_image(index) (:84-87) reads `reward = self.current_s[index, 6]` as a VIEW, overwrites the whole row
                    # with sample_s() (whose slot 6 is 0) and writes the view back: the accumulated reward is lost.
                    # Replicated as computed (SURVEY appendix C policy); verified against the reference by make_golden_env.py.
                    for k in range(6):
                        s[e, k] = env_randint(seed, game_offset + e, stage, k)
                    s[e, 6] = 0.0
                    changed[e] = True
                    break
            elif pi == 1:
                if s[e, 5] > 0:
                    s[e, 5] -= 1.0
            elif pi == 2:
                if s[e, 4] < 31:
                    s[e, 4] += 1.0
            elif pi == 3:
                if s[e, 4] > 0:
                    s[e, 4] -= 1.0
            else:
                raise ValueError('Invalid action')
    return changed


def render(s, last_r, imgs):
    """frames [n,64,64,1] float32 (HWC like the reference's current_frame_all)"""
    out = np.zeros((len(s), 64, 64, 1), dtype=np.float32)
    for e in range(len(s)):
        idx = int(np.dot(s[e, :6].astype(np.float32), BASES))
        img = imgs[idx].astype(np.float32).reshape(64, 64, 1).copy()
        r = last_r[e]
        if 0.0 <= r <= 1.0:
            img[0:3, 0:32] = r
        elif -1.0 <= r < 0.0:
            img[0:3, 32:64] = -r
        else:
            raise ValueError(f'Error: Reward: {r}')
        out[e] = img
    return out


def sprite_bank(n=3581):
    """deterministic stand-in for dsprites_ndarray_co1sh3sc6or40x32y32_64x64.npz (absent: .MISSING_LARGE_BLOBS):
    uint8 [n,64,64] binary images, one filled box per index."""
    imgs = np.zeros((n, 64, 64), dtype=np.uint8)
    for i in range(n):
        side = 6 + (i * 7) % 18
        y = 3 + (i * 13) % (61 - side)
        x = (i * 29) % (64 - side)
        imgs[i, y:y + side, x:x + side] = 1
    return imgs
