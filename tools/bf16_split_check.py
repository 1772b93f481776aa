"""dev tool (CPU): would a 3-term bf16 split of both operands (x = x1 + x2 + x3, 6 or 9 bf16 products, fp32 accumulate: what v_mfma_f32_32x32x16_bf16
would compute) hold the fp32 tolerances of the decoder contraction?  It does: the 6-product form is as accurate as a plain fp32 GEMM (VERDICT r3 item 9:
the numerical half of the experiment; no kernel was built -- the headline stays on exact fp32 MFMA)."""
import numpy as np, torch
torch.manual_seed(0)
def bf16(x):  # round-to-nearest-even to bf16, returned as float32
    return x.to(torch.bfloat16).to(torch.float32)
def split3(x):
    x1 = bf16(x); r = x - x1; x2 = bf16(r); x3 = bf16(r - x2)
    return x1, x2, x3
# a k_dec_a-like contraction: M = 64 out channels, K = 9 taps x 64 channels = 576, N = 1024 pixels; activations relu-like, weights ~ N(0, gain/sqrt(fan))
K, M, N = 576, 64, 4096
W = (torch.randn(M, K) * (1.15 / np.sqrt(K))).float()
X = torch.relu(torch.randn(K, N)).float() * 1.3
ref64 = (W.double() @ X.double())
ref32 = (W @ X)
w, x = split3(W), split3(X)
def mm32(a, b, chunk=16):
    # fp32 accumulation in chunks of K = 16 (one bf16 MFMA), products exact, chunk sums in fp32
    out = torch.zeros(a.shape[0], b.shape[1])
    for k0 in range(0, a.shape[1], chunk):
        out = out + a[:, k0:k0 + chunk] @ b[k0:k0 + chunk]
    return out
for terms, name in (([(0, 0)], '1 product (plain bf16)'), ([(0, 0), (0, 1), (1, 0)], '3 products'),
                    ([(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)], '6 products'), ([(i, j) for i in range(3) for j in range(3)], '9 products')):
    acc = torch.zeros(M, N)
    for (i, j) in sorted(terms, key=lambda t: -(t[0] + t[1])):      # small terms first
        acc = acc + mm32(w[i], x[j])
    err = (acc.double() - ref64).abs()
    e32 = (ref32.double() - ref64).abs()
    scale = ref64.abs().max().item()
    print(f'{name:24s} max abs err {err.max().item():.3e} (fp32 GEMM: {e32.max().item():.3e}), rel to max|y| {err.max().item() / scale:.3e}; '
          f'max rel err over |y| > 0.1: {(err / ref64.abs())[ref64.abs() > 0.1].max().item():.3e} (fp32: {(e32 / ref64.abs())[ref64.abs() > 0.1].max().item():.3e})')
