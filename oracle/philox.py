"""ORACLE (test infrastructure, never shipped on the product path).

Counter-based noise generator shared by the oracle and the HIP engine.

The reference draws its noise from torch's global generator (never seeded,
`/root/reference/src/torchmodel.py:54-56,130-132` reparameterize,
`nn.Dropout(0.5)` at `:44,47,50,96,99,102,109,112,115,118`).  That stream is
not addressable, so parity is defined in *injected-noise mode*: both sides
consume the Philox4x32-10 stream defined here (the HIP mirror lives in
`deep-active-inference-mc_amd/csrc/philox.h`), keyed by the logical identity
of every draw, so the result does not depend on batching or on GPU count.

Key      = (seed & 0xffffffff, seed >> 32)
Counter  = (blk | tag << 16, global_row, pass << 16 | sample, stage)

* dropout mask of a layer with F features: feature f of row r is KEPT iff
  bit (f & 31) of output word ((f >> 5) & 3) of counter blk = f >> 7 is 1
  (kept activations are scaled by 2.0 = 1/(1-p), p = 0.5).
* reparameterisation normals: element k of row r is Box-Muller lane (k & 3)
  of counter blk = k >> 2 (lanes 0,1 from words 0,1; lanes 2,3 from 2,3).
* categorical sampling uniform: word 0 of blk 0.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

# --- tags (which tensor the draw belongs to) --------------------------------
TAG_MID = 0x10   # + layer 0..2   ModelMid.ps_net dropouts   (torchmodel.py:44,47,50)
TAG_DEC = 0x20   # + layer 0..3   ModelDown.po_net dropouts  (torchmodel.py:109,112,115,118)
TAG_ENC = 0x30   # + layer 0..2   ModelDown.qs_net dropouts  (torchmodel.py:96,99,102)
TAG_EPS = 0x40   # reparameterisation normals                (torchmodel.py:55,131)
TAG_ACT = 0x50   # categorical action sampling uniform       (torchmodel.py:364,379)

# --- passes (which network evaluation inside one calculate_G stage) ---------
PASS_T1 = 0    # loop-1 transition_with_sample        (torchmodel.py:274)
PASS_D1 = 1    # loop-1 decoder                       (torchmodel.py:275)
PASS_E1 = 2    # loop-1 encoder_with_sample           (torchmodel.py:276)
PASS_T2 = 3    # loop-2 transition_with_sample        (torchmodel.py:288)
PASS_D2A = 4   # loop-2 decoder of the new transition (torchmodel.py:288)
PASS_D2B = 5   # loop-2 reparameterize + decoder      (torchmodel.py:291)
PASS_ROOT = 6  # root encoder + reparameterize        (torchmodel.py:228-229, mcts.py:158)
PASS_HABIT = 7 # habit net action sampling            (torchmodel.py:363-364,379)
PASS_SIM = 8   # mcts_step_simulate transitions       (torchmodel.py:368,382)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al. 2011). All args broadcastable uint32 arrays.
    Returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64); c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64); c3 = np.asarray(c3, dtype=np.uint64)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0            # < 2^64, exact in uint64
        p1 = M1 * c2
        hi0 = p0 >> np.uint64(32); lo0 = p0 & MASK32
        hi1 = p1 >> np.uint64(32); lo1 = p1 & MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def _key(seed):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed & 0xFFFFFFFF, seed >> 32


def stream_id(pas, sample):
    return ((int(pas) & 0xFFFF) << 16) | (int(sample) & 0xFFFF)


def dropout_mask(seed, tag, rows, n_feat, pas, sample, stage, row_offset=0):
    """float32 [rows, n_feat] of {0., 2.}: the x*mask*2 multiplier of nn.Dropout(0.5)
     for global rows row_offset..row_offset+rows-1."""
    k0, k1 = _key(seed)
    nblk = (n_feat + 127) // 128
    blk = np.arange(nblk, dtype=np.uint64)[None, :]
    row = (np.arange(rows, dtype=np.uint64) + np.uint64(row_offset))[:, None]
    w = philox4x32_10(blk | (np.uint64(tag) << np.uint64(16)), row,
                      np.uint64(stream_id(pas, sample)), np.uint64(stage), k0, k1)
    words = np.stack(w, axis=-1)                      # [rows, nblk, 4]
    bits = (words[..., None] >> np.arange(32, dtype=np.uint32)) & np.uint32(1)  # [rows,nblk,4,32]
    bits = bits.reshape(rows, nblk * 128)[:, :n_feat]
    return bits.astype(np.float32) * np.float32(2.0)


def _u01(x):
    # 24-bit uniform in (0,1), exactly representable in fp32
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def normals(seed, rows, n, pas, sample, stage, row_offset=0, tag=TAG_EPS):
    """float32 [rows, n] standard normals (Box-Muller in fp32)."""
    k0, k1 = _key(seed)
    nblk = (n + 3) // 4
    blk = np.arange(nblk, dtype=np.uint64)[None, :]
    row = (np.arange(rows, dtype=np.uint64) + np.uint64(row_offset))[:, None]
    x0, x1, x2, x3 = philox4x32_10(blk | (np.uint64(tag) << np.uint64(16)), row,
                                   np.uint64(stream_id(pas, sample)), np.uint64(stage), k0, k1)
    two_pi = np.float32(6.283185307179586)
    out = np.empty((rows, nblk, 4), dtype=np.float32)
    for lane, (a, b) in enumerate(((x0, x1), (x2, x3))):
        u1 = _u01(a); u2 = _u01(b)
        r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
        th = (two_pi * u2).astype(np.float32)
        out[..., 2 * lane] = r * np.cos(th).astype(np.float32)
        out[..., 2 * lane + 1] = r * np.sin(th).astype(np.float32)
    return out.reshape(rows, nblk * 4)[:, :n].copy()


def uniforms(seed, rows, pas, sample, stage, row_offset=0, tag=TAG_ACT):
    """float32 [rows] uniforms in (0,1) for inverse-CDF categorical sampling."""
    k0, k1 = _key(seed)
    row = np.arange(rows, dtype=np.uint64) + np.uint64(row_offset)
    x0, _, _, _ = philox4x32_10(np.uint64(tag) << np.uint64(16), row,
                                np.uint64(stream_id(pas, sample)), np.uint64(stage), k0, k1)
    return _u01(x0)


def uniform_fill(seed, shape, stream, lo=-1.0, hi=1.0):
    """Deterministic float32 uniform tensor for synthetic weights/frames
    (test-only; counter = (index, 0, stream, 0xFFFF0000))."""
    n = int(np.prod(shape))
    k0, k1 = _key(seed)
    nblk = (n + 3) // 4
    idx = np.arange(nblk, dtype=np.uint64)
    w = philox4x32_10(idx & MASK32, idx >> np.uint64(32), np.uint64(stream), np.uint64(0xFFFF0000), k0, k1)
    u = np.stack([_u01(x) for x in w], axis=-1).reshape(-1)[:n]
    return (np.float32(lo) + (np.float32(hi) - np.float32(lo)) * u).astype(np.float32).reshape(shape)
