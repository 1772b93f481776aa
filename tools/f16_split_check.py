"""dev tool (CPU): the operand splits of csrc/bf16x3.hip against an fp64 contraction -- three bf16 planes / six products (mfma_bf16x3) and
two fp16 planes / three products with power-of-two weight scaling (mfma_f16x2) -- on a decoder-like contraction (K = 576, weights ~ He-uniform,
ReLU activations over four decades of magnitude).  Products are formed exactly and summed in fp64 here, so the numbers are the OPERAND
REPRESENTATION error alone; the fp32 accumulation of the matrix pipe adds what a plain fp32 GEMM has (last line)."""
import numpy as np
rng = np.random.default_rng(0)
K, M, N = 576, 256, 512
W = (rng.uniform(-1, 1, (M, K)) * np.sqrt(6 / K) * 1.15).astype(np.float32)
X = np.maximum(rng.normal(0, 1, (K, N)), 0).astype(np.float32) * rng.choice([0.01, 0.1, 1, 5], (K, 1)).astype(np.float32)
ref = W.astype(np.float64) @ X.astype(np.float64)


def bf(x):
    u = x.astype(np.float32).view(np.uint32)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.view(np.float32)


def split3(a):
    hi = bf(a); r = a - hi; mid = bf(r); lo = bf(r - mid)
    return hi.astype(np.float64), mid.astype(np.float64), lo.astype(np.float64)


def split2(a, s):
    a = a * np.float32(s)
    hi = a.astype(np.float16); lo = (a - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


wh, wm, wl = split3(W); xh, xm, xl = split3(X)
b3 = wl @ xh + wh @ xl + wm @ xm + wm @ xh + wh @ xm + wh @ xh
print(f'max |y| {np.abs(ref).max():.3f}')
print(f'bf16 x 3, six products            max abs err {np.abs(b3 - ref).max():.3e}')
mx = np.abs(W).max()
sc = 2.0 ** (14 - np.frexp(mx)[1])
for sw, label in ((1.0, 'weights unscaled (low planes denormal)'), (sc, f'weights x 2^{int(np.log2(sc))} (the engine: largest |w| just below 2^14)')):
    wh2, wl2 = split2(W, sw); xh2, xl2 = split2(X, 1.0)
    h2 = (wl2 @ xh2 + wh2 @ xl2 + wh2 @ xh2) / sw
    print(f'fp16 x 2, three products, {label:70s} max abs err {np.abs(h2 - ref).max():.3e}')
print(f'plain fp32 GEMM (numpy)           max abs err {np.abs((W @ X).astype(np.float64) - ref).max():.3e}')
