// Fused decoder kernels for gfx950: the three ConvTranspose2d + ReLU layers, the final
// ConvTranspose2d(32,1) + Sigmoid and the per-image EFE reductions of
// /root/reference/src/torchmodel.py:120-127 (po_net.13..20), torchutils.py:26-37, torchmodel.py:210-212,289,292.
//
// One workgroup (4 waves) owns one decoder image; activations live in LDS between layers:
//
//   k_dec_a :  x4[16x16x64] --LDS--> ConvT(64,64,s1)+ReLU --LDS (in place)--> ConvT(64,64,s2)+ReLU --> y2[32x32x64] (HBM)
//   k_dec_b :  y2 strips --LDS--> ConvT(64,32,s2)+ReLU (registers) --MFMA--> 9 tap planes of the 32->1 conv
//              --LDS ring--> 3x3 gather + sigmoid + entropy / reward reduction (+ optional image store)
//
// LDS images are [pixel][64 ch] with the 16-byte channel-quad index XOR-swizzled by (pixel & 15), so the
// ds_read_b128 of an MFMA B fragment (32 pixels x same quad) is bank-conflict free without padding
// (cdna_hip_programming.md T2).  Weights are read as pre-packed A fragments straight from L2 (1 KiB coalesced
// per wave-load, shared by all workgroups).  fp32 MFMA (v_mfma_f32_32x32x2_f32): exact fp32 numerics.
#include "kernels.h"

namespace efe {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA4(ACC, AV, BV)                                                      \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).x, (BV).x, ACC, 0, 0, 0);   \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).y, (BV).y, ACC, 0, 0, 0);   \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).z, (BV).z, ACC, 0, 0, 0);   \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).w, (BV).w, ACC, 0, 0, 0);

__device__ __forceinline__ int swz(int pix, int c4) { return pix * 16 + (c4 ^ (pix & 15)); }   // float4 index

// ---------------------------------------------------------------------------------------------------------
// k_dec_a: ConvTranspose2d(64,64,3,s1,p1)+ReLU then ConvTranspose2d(64,64,3,s2,p1,op1)+ReLU, one image per WG.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) k_dec_a(const DecAArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 sm[];        // [257 pixels][16 quads]; pixel 256 = zeros
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int img = blockIdx.x;

    {   // stage the 16x16x64 input image (64 KiB), fully coalesced
        const float4* X = reinterpret_cast<const float4*>(a.x4) + (size_t)img * 4096;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = it * 256 + tid;
            const int pix = idx >> 4, c4 = idx & 15;
            sm[swz(pix, c4)] = X[idx];
        }
        if (tid < 16) sm[256 * 16 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    // this wave's 64 pixels / input positions: image rows 4w .. 4w+3, two 32-pixel tiles of two rows each
    int prow[2], pcol;
    pcol = j & 15;
    prow[0] = 4 * w + (j >> 4);
    prow[1] = 4 * w + 2 + (j >> 4);

    f32x16 acc[2][2];
    // ---------------- layer 1: out[oh,ow] = sum_{kh,kw} in[oh+1-kh, ow+1-kw] . W[:, :, kh, kw] ------------------
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
    {
        const float4* W = reinterpret_cast<const float4*>(a.w1) + lane;
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t - kh * 3;
            int base[2], sw[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int sy = prow[nt] + 1 - kh, sx = pcol + 1 - kw;
                const bool ok = sy >= 0 && sy < 16 && sx >= 0 && sx < 16;
                const int sp = ok ? sy * 16 + sx : 256;
                base[nt] = sp * 16; sw[nt] = sp & 15;
            }
            const float4* wt = W + (size_t)(t * 2) * 8 * 64;             // [tap][mtile 0..1][kc 0..7][lane]
#pragma unroll 2
            for (int kc = 0; kc < 8; ++kc) {
                const float4 a0 = wt[(0 * 8 + kc) * 64], a1 = wt[(1 * 8 + kc) * 64];
                const float4 b0 = sm[base[0] + ((2 * kc + h) ^ sw[0])];
                const float4 b1 = sm[base[1] + ((2 * kc + h) ^ sw[1])];
                MFMA4(acc[0][0], a0, b0) MFMA4(acc[0][1], a0, b1)
                MFMA4(acc[1][0], a1, b0) MFMA4(acc[1][1], a1, b1)
            }
        }
    }
    __syncthreads();            // every wave is done reading the input image
    // bias + ReLU, written back IN PLACE as the input image of layer 2 (same swizzled layout)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int pix = 64 * w + 32 * nt + j;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int c4 = mt * 8 + 2 * g4 + h;
                const float4 bb = reinterpret_cast<const float4*>(a.b1)[c4];
                float4 v;
                v.x = fmaxf(acc[mt][nt][4 * g4 + 0] + bb.x, 0.f); v.y = fmaxf(acc[mt][nt][4 * g4 + 1] + bb.y, 0.f);
                v.z = fmaxf(acc[mt][nt][4 * g4 + 2] + bb.z, 0.f); v.w = fmaxf(acc[mt][nt][4 * g4 + 3] + bb.w, 0.f);
                sm[swz(pix, c4)] = v;
            }
    }
    __syncthreads();

    // ---------------- layer 2 (stride 2): 4 output parities, oh = 2*ih - 1 + kh --------------------------------
    float* Y = a.y2 + (size_t)img * (32 * 32 * 64);
    const float4* W2 = reinterpret_cast<const float4*>(a.w2) + lane;
    for (int par = 0; par < 4; ++par) {
        const int ph = par >> 1, pw = par & 1;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        const int ntaps = (1 + ph) * (1 + pw);
        for (int t = 0; t < ntaps; ++t) {
            const int th = t / (1 + pw), tw = t - th * (1 + pw);
            const int kh = ph ? (th ? 2 : 0) : 1, da = (ph && th == 0) ? 1 : 0;
            const int kw = pw ? (tw ? 2 : 0) : 1, db = (pw && tw == 0) ? 1 : 0;
            int base[2], sw[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int sy = prow[nt] + da, sx = pcol + db;
                const bool ok = sy < 16 && sx < 16;
                const int sp = ok ? sy * 16 + sx : 256;
                base[nt] = sp * 16; sw[nt] = sp & 15;
            }
            const float4* wt = W2 + (size_t)((kh * 3 + kw) * 2) * 8 * 64;
#pragma unroll 2
            for (int kc = 0; kc < 8; ++kc) {
                const float4 a0 = wt[(0 * 8 + kc) * 64], a1 = wt[(1 * 8 + kc) * 64];
                const float4 b0 = sm[base[0] + ((2 * kc + h) ^ sw[0])];
                const float4 b1 = sm[base[1] + ((2 * kc + h) ^ sw[1])];
                MFMA4(acc[0][0], a0, b0) MFMA4(acc[0][1], a0, b1)
                MFMA4(acc[1][0], a1, b0) MFMA4(acc[1][1], a1, b1)
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float* yp = Y + ((size_t)(2 * prow[nt] + ph) * 32 + (2 * pcol + pw)) * 64;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c4 = mt * 8 + 2 * g4 + h;
                    const float4 bb = reinterpret_cast<const float4*>(a.b2)[c4];
                    float4 v;
                    v.x = fmaxf(acc[mt][nt][4 * g4 + 0] + bb.x, 0.f); v.y = fmaxf(acc[mt][nt][4 * g4 + 1] + bb.y, 0.f);
                    v.z = fmaxf(acc[mt][nt][4 * g4 + 2] + bb.z, 0.f); v.w = fmaxf(acc[mt][nt][4 * g4 + 3] + bb.w, 0.f);
                    reinterpret_cast<float4*>(yp)[c4] = v;
                }
        }
    }
}

void launch_dec_a(const DecAArgs& a, hipStream_t st) {
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute((const void*)k_dec_a, hipFuncAttributeMaxDynamicSharedMemorySize, 257 * 16 * sizeof(float4)); once = true; }
    hipLaunchKernelGGL(k_dec_a, dim3(a.rows), dim3(256), 257 * 16 * sizeof(float4), st, a);
}

// ---------------------------------------------------------------------------------------------------------
// k_dec_b: ConvTranspose2d(64,32,3,s2,p1,op1)+ReLU, ConvTranspose2d(32,1,3,s1,p1)+Sigmoid and the per-image
// reduction, one image per WG, 8 strips of 4 input rows (8 output rows).
//
// The 32->1 conv is applied to the layer-3 accumulators while they are still in registers: for each finished
// 32(co) x 32(pixel) tile, 16 extra MFMAs contract the channel axis against the 9 taps,
//     T[tap][q] = sum_co W4[co][tap] * relu(y3[q][co] + b3[co]),
// using the accumulator register e of every lane directly as the B operand (lane (q,h) holds
// co = (e&3) + 8*(e>>2) + 4h, which is exactly the k-pair the f32 MFMA expects).  The 9 planes go to an LDS ring
// and the 3x3 "gather"  out[oh,ow] = b4 + sum_{kh,kw} T[kh*3+kw][oh+1-kh][ow+1-kw]  is done once the rows
// above and below exist.  y3 (512 KiB per image) never exists in memory.
// ---------------------------------------------------------------------------------------------------------
constexpr int DB_ZERO = 160;                 // zero pixel slot of the 5-row input strip
constexpr int DB_IN_F4 = (160 + 1) * 16;     // float4s
constexpr int DB_YROWS = 10;

__global__ void __launch_bounds__(256, 2) k_dec_b(const DecBArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 sm[];        // input strip, then T ring
    float* sT = reinterpret_cast<float*>(sm + DB_IN_F4);               // [10 rows][9 taps][64 cols]
    __shared__ float sred[4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int rp = w >> 1, pg = w & 1;
    const int img = blockIdx.x;

    const int mg = a.m0 + img;
    const int g = mg / a.rows_per_group;
    const int r = mg - g * a.rows_per_group;
    int gt, gp, gs;
    group_decode(a.gm, g, gt, gp, gs);
    const int mode = (gp == 0 && a.reward0) ? 1 : 0;
    const int slot = (gp == 0 && a.store0) ? gt * a.gm.S + gs : -1;
    float* po = (slot >= 0) ? a.po + ((size_t)slot * a.rows_per_group + r) * 4096 : nullptr;

    // per-lane constants: layer-3 bias of the 16 channels this lane holds, and the A fragments of the 32->1 conv
    float b3[16], w4f[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int co = (e & 3) + 8 * (e >> 2) + 4 * h;
        b3[e] = a.b3[co];
        w4f[e] = (j < 9) ? a.w4[j * 32 + co] : 0.f;                    // A[i = tap][k = h]
    }
    if (tid < 16) sm[DB_ZERO * 16 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);

    const float4* X = reinterpret_cast<const float4*>(a.y2) + (size_t)img * (32 * 32 * 16);
    const float4* W3 = reinterpret_cast<const float4*>(a.w3) + lane;
    const float D1 = 1.00001f, D0 = 0.00001f;      // fp32 constants of log_bernoulli / entropy_bernoulli
    float part = 0.f;

    for (int s = 0; s < 8; ++s) {
        // ---- stage input rows 4s .. 4s+4 (row 32 does not exist: zeros)
#pragma unroll
        for (int it = 0; it < 10; ++it) {
            const int idx = it * 256 + tid;                            // 0..2559 = 5 rows x 32 px x 16 quads
            const int rl = idx >> 9, rem = idx & 511, px = rem >> 4, c4 = rem & 15;
            const int grow = 4 * s + rl;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (grow < 32) v = X[((size_t)grow * 32 + px) * 16 + c4];
            sm[swz(rl * 32 + px, c4)] = v;
        }
        __syncthreads();

        // ---- MFMA phase: wave (rp, pg) owns local rows 2rp, 2rp+1 and two of the four output parities
#pragma unroll 1
        for (int pi = 0; pi < 2; ++pi) {
            const int par = pg ? (pi ? 2 : 1) : (pi ? 0 : 3);          // pg0: (1,1),(0,0)   pg1: (0,1),(1,0)
            const int ph = par >> 1, pw = par & 1;
            f32x16 acc[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
            const int ntaps = (1 + ph) * (1 + pw);
            for (int t = 0; t < ntaps; ++t) {
                const int th = t / (1 + pw), tw = t - th * (1 + pw);
                const int kh = ph ? (th ? 2 : 0) : 1, da = (ph && th == 0) ? 1 : 0;
                const int kw = pw ? (tw ? 2 : 0) : 1, db = (pw && tw == 0) ? 1 : 0;
                int base[2], sw[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int sx = j + db;
                    const int sp = (sx < 32) ? (2 * rp + nt + da) * 32 + sx : DB_ZERO;
                    base[nt] = sp * 16; sw[nt] = sp & 15;
                }
                const float4* wt = W3 + (size_t)(kh * 3 + kw) * 8 * 64;      // [tap][kc][lane], one 32-channel tile
#pragma unroll 4
                for (int kc = 0; kc < 8; ++kc) {
                    const float4 a0 = wt[kc * 64];
                    const float4 b0 = sm[base[0] + ((2 * kc + h) ^ sw[0])];
                    const float4 b1 = sm[base[1] + ((2 * kc + h) ^ sw[1])];
                    MFMA4(acc[0], a0, b0) MFMA4(acc[1], a0, b1)
                }
            }
            // ---- bias + ReLU in registers, then contract channels against the 9 taps of the final conv
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x16 T;
#pragma unroll
                for (int e = 0; e < 16; ++e) T[e] = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    T = __builtin_amdgcn_mfma_f32_32x32x2f32(w4f[e], fmaxf(acc[nt][e] + b3[e], 0.f), T, 0, 0, 0);
                // lane (q = j, h) now holds taps 4h + (0..3) in T[0..3] and tap 8 in T[4] (h == 0 only)
                const int orow = 2 * (4 * s + 2 * rp + nt) + ph, ocol = 2 * j + pw;
                float* tp = sT + ((orow % DB_YROWS) * 9) * 64 + ocol;
                tp[(4 * h + 0) * 64] = T[0]; tp[(4 * h + 1) * 64] = T[1];
                tp[(4 * h + 2) * 64] = T[2]; tp[(4 * h + 3) * 64] = T[3];
                if (h == 0) tp[8 * 64] = T[4];
            }
        }
        __syncthreads();

        // ---- gather: output rows 8s-1 .. 8s+6 are complete now (row 63 after the last strip)
        const int nq = (s == 7) ? 3 : 2;
        for (int q = 0; q < nq; ++q) {
            const int p = q * 256 + tid;
            if (q == 2 && tid >= 64) break;
            const int oh = 8 * s - 1 + (p >> 6), ow = p & 63;
            if (oh < 0) continue;
            float v = a.b4;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int tr = oh + 1 - kh;
                if (tr < 0 || tr > 63) continue;
                const float* trow = sT + ((tr % DB_YROWS) * 9 + kh * 3) * 64;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int tc = ow + 1 - kw;
                    if (tc >= 0 && tc < 64) v += trow[kw * 64 + tc];
                }
            }
            const float pr = 1.0f / (1.0f + expf(-v));
            if (po) po[oh * 64 + ow] = pr;
            if (mode == 0) part += -(1.0f - pr) * logf(D1 - pr) - pr * logf(D0 + pr);
            else           // target = 1 for image rows h < 32, 0 below (NCHW broadcast of the port)
                part += (oh < 32) ? pr * logf(D1) + (1.0f - pr) * logf(D1 - 1.0f) : pr * logf(D0) + (1.0f - pr) * logf(D1);
        }
        // no barrier here: the next strip's staging only touches the input buffer (all waves are past the MFMA
        // phase), and its T writes come after the next barrier, i.e. after every thread finished this gather.
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) sred[w] = part;
    __syncthreads();
    if (tid == 0) a.val[mg] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

void launch_dec_b(const DecBArgs& a, hipStream_t st) {
    const size_t lds = DB_IN_F4 * sizeof(float4) + DB_YROWS * 9 * 64 * sizeof(float);
    hipLaunchKernelGGL(k_dec_b, dim3(a.rows), dim3(256), lds, st, a);
}

}  // namespace efe
