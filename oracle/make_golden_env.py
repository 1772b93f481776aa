"""ORACLE tooling (build container only): captures fixtures of the reference's Dynamic-dSprites `Game`
(/root/reference/src/game_environment.py) for SURVEY row 8f-3.  The dSprites archive is absent, so `numpy.load` is patched
to hand the Game the synthetic sprite bank of oracle/env_oracle.py (the shapes the Game reads: imgs uint8 [N,64,64],
metadata['latents_sizes'] = [1,3,6,40,32,32]); `torch.randint` / `torch.rand` are patched to consume the addressable
Philox stream (tag 0x60) in the reference's own call order.  Only tensors are written to tests/golden/env.npz.

Usage:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_env
"""
import os
import sys
import io
import contextlib
import numpy as np

sys.dont_write_bytecode = True
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import env_oracle as EV

REF = '/root/reference'
SEED = 77


def main():
    sys.path.insert(0, REF)
    bank = EV.sprite_bank()
    meta = np.empty((), dtype=object)
    meta[()] = {'latents_sizes': np.array([1, 3, 6, 40, 32, 32])}
    fake = {'imgs': bank, 'latents_values': np.zeros((1, 6)), 'latents_classes': np.zeros((1, 6), dtype=np.int64), 'metadata': meta}
    np_load = np.load
    np.load = lambda *a, **k: fake
    ctx = {'stage': 1000, 'game': 0, 'k': 0, 'rand_blk': 6}

    def randint(*args, **kw):
        if len(args) == 2:
            high, size = args
        else:
            _, high, size = args
        high = int(high)
        k = ctx['k']; ctx['k'] = (k + 1) % 6
        assert high == EV.SIZES[k], (high, k)
        if tuple(size) == (1,):
            return torch.tensor([EV.env_randint(SEED, ctx['game'], ctx['stage'], k)])
        return torch.tensor([EV.env_randint(SEED, g, ctx['stage'], k) for g in range(size[0])])

    def rand(*size, **kw):
        n = size[0]
        blk = ctx['rand_blk']; ctx['rand_blk'] = 7 if blk == 6 else 6
        return torch.tensor([float(EV.env_u(SEED, g, ctx['stage'], blk)) for g in range(n)], dtype=torch.float32)

    torch.randint = randint
    torch.rand = rand
    from src.game_environment import Game

    G, T, REPEATS = 6, 6, 2
    with contextlib.redirect_stdout(io.StringIO()):
        games = Game(G)                                   # __init__ draws (new_image_all) use stage 1000
        s_init, r_init = games.current_s.numpy().copy(), games.last_r.numpy().copy()
        assert np.array_equal(s_init, EV.new_image_all(SEED, np.zeros((G, 7), np.float32), 1000)) and not r_init.any(), 'constructor mismatch'
        ctx.update(stage=0, k=0, rand_blk=6)
        games.randomize_environment_all()
    np.load = np_load
    s_reset, r_reset = games.current_s.numpy().copy(), games.last_r.numpy().copy()
    o_s, o_r = EV.reset(SEED, G, 0)
    assert np.array_equal(s_reset, o_s) and np.array_equal(r_reset, o_r), 'reset mismatch'

    games.current_s[:, 5] = torch.tensor([31.0, 30.0, 5.0, 31.0, 0.0, 29.0])      # some games about to finish a round
    games.current_s[:, 4] = torch.tensor([3.0, 20.0, 31.0, 16.0, 0.0, 15.0])
    games.current_s[:, 1] = torch.tensor([0.0, 1.0, 2.0, 0.0, 1.0, 2.0])
    s_in, r_in = games.current_s.numpy().copy(), games.last_r.numpy().copy()
    frames_in = games.current_frame_all().numpy().copy()
    os_, or_ = s_in.copy(), r_in.copy()
    assert np.array_equal(EV.render(os_, or_, bank), frames_in), 'render mismatch'

    actions = np.array([[0, 0, 2, 0, 1, 0], [0, 0, 2, 3, 1, 0], [1, 0, 3, 0, 3, 0], [0, 2, 0, 0, 2, 3], [3, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0]])
    states, rs, frames, changed = [], [], [], []
    for t in range(T):
        ch = np.zeros(G, dtype=bool)
        for e in range(G):
            ctx.update(stage=1 + t, game=e, k=0)
            ch[e] = games.pi_to_action(int(actions[t, e]), e, repeats=REPEATS)
        och = EV.step(SEED, os_, or_, actions[t], REPEATS, 1 + t)
        if not (np.array_equal(games.current_s.numpy(), os_) and np.array_equal(games.last_r.numpy(), or_)):
            print('ref', games.current_s.numpy(), games.last_r.numpy()); print('orc', os_, or_)
            raise AssertionError(f'step {t} mismatch')
        assert np.array_equal(ch, och)
        f = games.current_frame_all().numpy().copy()
        assert np.array_equal(EV.render(os_, or_, bank), f)
        states.append(os_.copy()); rs.append(or_.copy()); frames.append(f[..., 0]); changed.append(ch)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'env.npz'), seed=SEED, s_init=s_init, r_init=r_init, init_stage=1000, s_reset=s_reset, r_reset=r_reset, s_in=s_in,
                        r_in=r_in, frames_in=frames_in[..., 0], actions=actions, repeats=REPEATS, states=np.stack(states),
                        last_r=np.stack(rs), frames=np.stack(frames), changed=np.stack(changed))
    print('env golden written; rounds finished per step:', [int(c.sum()) for c in changed])


if __name__ == '__main__':
    main()
