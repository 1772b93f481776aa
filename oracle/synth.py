"""ORACLE / test infrastructure: deterministic synthetic weights and frames.

No trained checkpoint or dSprites file ships with the reference
(`/root/reference/.MISSING_LARGE_BLOBS`), so tests and the bench synthesise
weights of the reference architecture (`/root/reference/src/torchmodel.py:10-128`,
state_dict key names of SURVEY.md section 8b) from the Philox stream.
Bounds are He-uniform-like with a gain chosen so the decoder leaves the
p = 0.5 plateau and G depends visibly on (s, pi).
"""
import numpy as np
from . import philox as PX

# (key, shape, fan_in)
SPECS = [
    ('top.qpi_net.0', (128, 10)), ('top.qpi_net.2', (128, 128)), ('top.qpi_net.4', (4, 128)),
    ('mid.ps_net.0', (512, 14)), ('mid.ps_net.3', (512, 512)), ('mid.ps_net.6', (512, 512)), ('mid.ps_net.9', (20, 512)),
    ('down.qs_net.0', (32, 1, 3, 3)), ('down.qs_net.2', (32, 32, 3, 3)), ('down.qs_net.4', (64, 32, 3, 3)),
    ('down.qs_net.6', (64, 64, 3, 3)), ('down.qs_net.9', (256, 576)), ('down.qs_net.12', (256, 256)),
    ('down.qs_net.15', (256, 256)), ('down.qs_net.18', (20, 256)),
    ('down.po_net.0', (256, 10)), ('down.po_net.3', (256, 256)), ('down.po_net.6', (256, 256)),
    ('down.po_net.9', (16384, 256)),
    # ConvTranspose2d weights are [Cin, Cout, kh, kw]
    ('down.po_net.13', (64, 64, 3, 3)), ('down.po_net.15', (64, 64, 3, 3)),
    ('down.po_net.17', (64, 32, 3, 3)), ('down.po_net.19', (32, 1, 3, 3)),
]
CONVT = {'down.po_net.13': 1, 'down.po_net.15': 2, 'down.po_net.17': 2, 'down.po_net.19': 1}  # stride


def _fan_in(name, shape):
    if len(shape) == 2:
        return shape[1]
    if name in CONVT:      # transposed conv: each output sees ~ Cin*9/stride^2 inputs
        return shape[0] * 9 / (CONVT[name] ** 2)
    return shape[1] * 9


def specs_for(pi_dim=4, channels=1, resolution=64):
    """(key, shape) list for a geometry: SPECS for the reference's dSprites model; build-defined sizes otherwise"""
    if (pi_dim, channels, resolution) == (4, 1, 64):
        return SPECS
    h = resolution
    for _ in range(4):
        h = (h - 3) // 2 + 1
    base = resolution // 2 if resolution == 32 else resolution // 4
    repl = {'top.qpi_net.4': (pi_dim, 128), 'mid.ps_net.0': (512, pi_dim + 10), 'down.qs_net.0': (32, channels, 3, 3),
            'down.qs_net.9': (256, 64 * h * h), 'down.po_net.9': (64 * base * base, 256), 'down.po_net.19': (32, channels, 3, 3)}
    return [(k, repl.get(k, shp)) for k, shp in SPECS]


def make_weights(seed=1234, gain=1.0, pi_dim=4, channels=1, resolution=64):
    """dict key -> float32 array; keys are '<top|mid|down>.<state_dict key>'."""
    w = {}
    for i, (name, shape) in enumerate(specs_for(pi_dim, channels, resolution)):
        fan = _fan_in(name, shape)
        # dropout(0.5) doubles the second moment of kept activations, so use sqrt(3/fan) for the
        # layers that follow a dropout and sqrt(6/fan) elsewhere; overall scale via `gain`.
        g = 1.0 if name.startswith('mid.') else gain      # transition net: keep the depth recursion contractive
        bound = g * np.sqrt(3.0 / fan)
        w[name + '.weight'] = PX.uniform_fill(seed, shape, 2 * i, -bound, bound)
        bshape = (shape[1],) if name in CONVT else (shape[0],)
        w[name + '.bias'] = PX.uniform_fill(seed, bshape, 2 * i + 1, -0.1, 0.1)
    w['mid.ps_net.9.weight'] *= 0.3   # bounded imagined states over >= 7 chained stages
    # final logvar halves: keep them moderate so exp(0.5*logvar) stays O(0.3)
    for k in ('mid.ps_net.9', 'down.qs_net.18'):
        w[k + '.bias'][10:] -= 2.0
    return w


def make_frames_rgb(seed, n, channels=3, resolution=84):
    """[n, C, R, R] float32 synthetic colour frames (a filled box per channel on a dim background), values in [0, 1]"""
    u = PX.uniform_fill(seed, (n, channels, 5), 2000, 0.0, 1.0)
    frames = np.zeros((n, channels, resolution, resolution), dtype=np.float32)
    for i in range(n):
        for c in range(channels):
            side = 8 + int(u[i, c, 0] * (resolution // 3))
            y = int(u[i, c, 1] * (resolution - side)); x = int(u[i, c, 2] * (resolution - side))
            frames[i, c] = 0.1 * u[i, c, 3]
            frames[i, c, y:y + side, x:x + side] = 0.5 + 0.5 * u[i, c, 4]
    return frames


def make_frames(seed, n):
    """[n, 1, 64, 64] float32 dSprites-like frames: one filled square plus the reward bar of
    `/root/reference/src/game_environment.py:44-54,70-71` (rows 0..2, left half = +r, right half = -r)."""
    u = PX.uniform_fill(seed, (n, 4), 1000, 0.0, 1.0)
    frames = np.zeros((n, 1, 64, 64), dtype=np.float32)
    for i in range(n):
        side = 6 + int(u[i, 0] * 18)
        y = 3 + int(u[i, 1] * (61 - side))
        x = int(u[i, 2] * (64 - side))
        frames[i, 0, y:y + side, x:x + side] = 1.0
        r = 2.0 * u[i, 3] - 1.0
        if r > 0:
            frames[i, 0, 0:3, 0:32] = r
        else:
            frames[i, 0, 0:3, 32:64] = -r
    return frames
