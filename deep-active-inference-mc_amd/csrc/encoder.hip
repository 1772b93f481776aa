// Fused encoder trunk for gfx950: the four Conv2d(k3, stride 2, no padding) + ReLU layers of
// /root/reference/src/torchmodel.py:85-92 (qs_net.0..7), 64x64x1 -> 31x31x32 -> 15x15x32 -> 7x7x64 -> 3x3x64,
// one image per workgroup, every intermediate in LDS; only the 576-vector (NHWC p*64+c order, the column order the
// packed first dense layer expects) leaves the chip.
//
//   conv1 (Cin = 1) is never materialised: the conv1 activations a conv2 tap needs are produced by five MFMAs straight from
//   the 16 KiB input image in LDS, in the register layout of the conv2 B operand.  conv2, conv3 are MFMA tap contractions
//   (v_mfma_f32_32x32x2_f32) over padded LDS images, conv4 (9 pixels) runs on the 16-column form v_mfma_f32_16x16x4_f32.
#include "mfma_pipe.h"

#ifndef EFE_ENC_WAVES
#define EFE_ENC_WAVES 3      // waves per SIMD = workgroups per CU (50 KiB LDS, <= 168 VGPRs)
#endif

namespace efe {

constexpr int EN_IMG = 0;                      // [64][64] input image                       (floats)
// Activation images in LDS are PADDED, not swizzled: a pixel's channels take P2 = 9 (conv2 output, 8 quads + 1) / P3 = 17 (conv3 output,
// 16 + 1) float4 slots.  The stride-2 taps of 16 consecutive lanes then fall 2-way on the banks (the XOR swizzle: 4-way, its key only
// takes even values), the epilogue's stores are conflict-free, and a fragment address is base + constant: the chunk offset is an
// immediate of the ds_read instead of an XOR + add per chunk (conv3 / conv4 spent more VALU than MFMA instructions on those).
constexpr int EN_P2 = 9, EN_P3 = 17;
constexpr int EN_C2 = 4096;                    // [225 px][EN_P2 quads] conv2 output
constexpr int EN_C3 = EN_IMG;                  // [49 px][EN_P3 quads] conv3 output: aliases the input image (dead once conv2 is done);
                                               // 49 * 68 = 3332 <= 4096 floats
constexpr int EN_ONE = EN_C2 + 225 * 4 * EN_P2;       // 320 x 1.0f: the "pixel" the conv1 bias row multiplies (any tap offset stays inside)
constexpr int EN_B2 = EN_ONE + 320;             // conv2 / conv3 / conv4 biases [32] [64] [64]: read in the epilogues from LDS (a global
constexpr int EN_B3 = EN_B2 + 32;              // load there exposes an L2 round trip per phase and image)
constexpr int EN_B4 = EN_B3 + 64;
constexpr int EN_END = EN_B4 + 64;             // 12676 floats = 50704 B: three workgroups per CU (160 KiB)

__global__ void __launch_bounds__(256, EFE_ENC_WAVES) k_enc_trunk(const EncArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    float4* sm4 = reinterpret_cast<float4*>(smf);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid; i < 320; i += 256) smf[EN_ONE + i] = 1.0f;
    // conv1 as a [32 ch] x [K = 9 taps + bias] A operand: lane (ch, h) holds column 2 s + h of MFMA step s, for the whole kernel
    float a1[5];
#pragma unroll
    for (int s_ = 0; s_ < 5; ++s_) { const int k = 2 * s_ + (lane >> 5); a1[s_] = k < 9 ? a.w1[k * 32 + (lane & 31)] : a.b1[lane & 31]; }
    if (tid < 160) smf[EN_B2 + tid] = tid < 32 ? a.b2[tid] : tid < 96 ? a.b3[tid - 32] : a.b4[tid - 96];
    const float4* W3 = reinterpret_cast<const float4*>(a.w3);             // [9][2][4][64] (uniform base, see TapPipe)
    const float4* W4 = reinterpret_cast<const float4*>(a.w4);             // [9][4][4][64], packed for the 16x16x4 form

    for (int img = blockIdx.x; img < a.rows; img += gridDim.x) {
        if (!row_live(a.live, img)) continue;            // a dead row of the call (efe_rows.mask): workgroup-uniform
        // lane index laundered per image: hoisted out of the image loop, the per-phase LDS addresses derived from it are ~20 spilled VGPRs
        int ll = lane; asm volatile("" : "+v"(ll));
        const int j = ll & 31, h = ll >> 5;
        {   // stage the image (16 KiB)
            const f32x4* X = reinterpret_cast<const f32x4*>(a.o) + (size_t)img * 1024;
            f32x4* d = reinterpret_cast<f32x4*>(smf + EN_IMG);
#pragma unroll
            for (int it = 0; it < 4; ++it) d[it * 256 + tid] = X[it * 256 + tid];
        }
        __syncthreads();

        // ================= conv2 with conv1 on the matrix pipe ==================
        // 8 tiles of two conv2 rows x 16 columns (15 live): lane (h, r, ox) <-> conv2 pixel (2 T + r, ox).  The B operand of conv2 tap
        // (kh, kw) is conv1 + ReLU at position (2 oy + kh, 2 ox + kw), all 32 channels: five 32x32x2 MFMAs over K = 9 image taps + the
        // bias against a constant 1 give it in exactly the register layout the conv2 MFMAs consume (D row 8 g + 4 h + r = B channel of
        // chunk g), so no conv1 value ever goes through LDS or the VALU (a VALU instruction costs ~19 cycles of wave time next to a
        // streaming MFMA wave: the 9-FMA-per-value form spent as long there as in the conv2 MFMAs).  Tap kw = 2 of a pixel is tap kw = 0
        // of its right neighbour: one DPP row shift per register instead of five MFMAs.
        {
            const int rr = j >> 4, ox = j & 15;
            bool pv[2]; int m[2];
            const float* pa[2][5];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int oy = 2 * (w + 4 * nt) + rr;
                pv[nt] = ox < 15 && oy < 15;
                const int oyc = oy < 15 ? oy : 14;                        // the half tile below the image recomputes row 14 (discarded)
                m[nt] = oyc * 15 + ox;
#pragma unroll
                for (int s_ = 0; s_ < 5; ++s_) {
                    const int k = 2 * s_ + h;                             // this lane's conv1 tap of MFMA step s_ (k = 9: the bias row)
                    pa[nt][s_] = (k < 9) ? smf + EN_IMG + (4 * oyc + k / 3) * 64 + 4 * ox + k % 3 : smf + EN_ONE;
                }
            }
            f32x16 acc[2], bA[2], bB[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
            float pb[2][5];
            auto patch = [&](int kh, int kw) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int s_ = 0; s_ < 5; ++s_) pb[nt][s_] = pa[nt][s_][(2 * kh) * 64 + 2 * kw];
            };
            auto conv1 = [&](f32x16 (&bx)[2]) {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                bx[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], pb[0][0], z, 0, 0, 0);
                bx[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], pb[1][0], z, 0, 0, 0);
#pragma unroll
                for (int s_ = 1; s_ < 5; ++s_) {
                    bx[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s_], pb[0][s_], bx[0], 0, 0, 0);
                    bx[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s_], pb[1][s_], bx[1], 0, 0, 0);
                }
            };
            auto relu = [&](f32x16 (&bx)[2]) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) bx[nt][e] = relu_bits(bx[nt][e]);
            };
            auto shl = [&](f32x16 (&bx)[2], int kc) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int e = 4 * kc; e < 4 * kc + 4; ++e) bx[nt][e] = dpp_shl1_zero(bx[nt][e]);
            };
            const unsigned ln2 = (unsigned)lane * 16u;
            const __amdgpu_buffer_rsrc_t wr2 = wrsrc(a.w2);              // [9 taps][4 chunks][64 lanes] float4
            float4 an = wfrag(wr2, ln2, 0);
            // conv2 tap `tap` over the 32 conv1 channels in bx; `nxt` = the tap whose first weight fragment is requested at the end
            auto conv2 = [&](int tap, int nxt, f32x16 (&bx)[2], auto&& during) {
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const float4 av = an;
                    an = wfrag(wr2, ln2, (size_t)(kc < 3 ? tap * 4 + kc + 1 : nxt * 4) * 64);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const float4 b4 = make_float4(bx[nt][4 * kc], bx[nt][4 * kc + 1], bx[nt][4 * kc + 2], bx[nt][4 * kc + 3]);
                        MFMA4(acc[nt], av, b4)
                    }
                    during(kc);
                }
            };
            patch(0, 0); conv1(bA); relu(bA); patch(0, 1);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                // tap (kh, 0) from bA; under it conv1 of (kh, 1) -> bB
                conv2(kh * 3 + 0, kh * 3 + 2, bA, [&](int kc) {
                    if (kc == 0) conv1(bB);
                    if (kc == 2) relu(bB);
                    if (kc == 3) shl(bA, 0);
                });
                // tap (kh, 2): bA shifted one pixel to the left, a chunk ahead of its use
                conv2(kh * 3 + 2, kh * 3 + 1, bA, [&](int kc) { if (kc < 3) shl(bA, kc + 1); else if (kh < 2) patch(kh + 1, 0); });
                // tap (kh, 1) from bB; under it conv1 of (kh + 1, 0) -> bA
                conv2(kh * 3 + 1, kh < 2 ? kh * 3 + 3 : 8, bB, [&](int kc) {
                    if (kh < 2) {
                        if (kc == 0) conv1(bA);
                        if (kc == 2) relu(bA);
                        if (kc == 3) patch(kh + 1, 1);
                    }
                });
            }
            // bias + ReLU -> conv2 image in LDS
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (!pv[nt]) continue;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c4 = 2 * g4 + h;
                    const float4 bb = sm4[EN_B2 / 4 + c4];
                    float4 v;
                    v.x = fmaxf(acc[nt][4 * g4 + 0] + bb.x, 0.f); v.y = fmaxf(acc[nt][4 * g4 + 1] + bb.y, 0.f);
                    v.z = fmaxf(acc[nt][4 * g4 + 2] + bb.z, 0.f); v.w = fmaxf(acc[nt][4 * g4 + 3] + bb.w, 0.f);
                    sm4[EN_C2 / 4 + m[nt] * EN_P2 + c4] = v;
                }
            }
        }
        __syncthreads();

        // ================= conv3: 49 pixels (2 tiles) x 64 channels (2 tiles): one (mt, nt) unit per wave ==================
        {
            const int mt = w & 1, nt = w >> 1;
            const int m = 32 * nt + j;
            const bool pv = m < 49;
            const int mm = pv ? m : 0;
            const int oy = mm / 7, ox = mm - oy * 7;
            f32x16 acc[1][1];
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.f;
            tap_loop_kc<1, 1, 4>(acc, 9, W3, sm4 + EN_C2 / 4, h, [&](int t, int (&bs)[1], int (&sw)[1], int& wt) {
                const int kh = t / 3, kw = t - kh * 3;
                wt = t;
                const int sp = (2 * oy + kh) * 15 + 2 * ox + kw;
                bs[0] = sp * EN_P2; sw[0] = 0;
            }, PackedWIdx{2, 4, mt});
            if (pv) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c4 = mt * 8 + 2 * g4 + h;
                    const float4 bb = sm4[EN_B3 / 4 + c4];
                    float4 v;
                    v.x = fmaxf(acc[0][0][4 * g4 + 0] + bb.x, 0.f); v.y = fmaxf(acc[0][0][4 * g4 + 1] + bb.y, 0.f);
                    v.z = fmaxf(acc[0][0][4 * g4 + 2] + bb.z, 0.f); v.w = fmaxf(acc[0][0][4 * g4 + 3] + bb.w, 0.f);
                    sm4[EN_C3 / 4 + m * EN_P3 + c4] = v;
                }
            }
        }
        __syncthreads();

        // ================= conv4: 9 pixels x 64 channels on v_mfma_f32_16x16x4_f32 (9 of 16 columns live, against 9 of 32 on the wide
        // tile): wave w owns channels 16w..16w+15 over the whole K = 9 taps x 64, two accumulator chains (the form's dependent latency
        // is 40 cycles against a 32-cycle issue).  D row 4q + r of lane (q, px) = four consecutive channels of one pixel: one store.
        {
            const int px = ll & 15, q = ll >> 4;
            const bool pv = px < 9;
            const int mm = pv ? px : 0;
            const int oy = mm / 3, ox = mm - oy * 3;
            const unsigned ln = (unsigned)lane * 16u;
            const __amdgpu_buffer_rsrc_t wr = wrsrc(W4);
            constexpr int PD = 4;                           // weight fragments (L2) requested four 128-cycle steps ahead
            auto bfrag = [&](int i) {
                const int t = i >> 2, kh = t / 3, kw = t - kh * 3;
                const int sp = (2 * oy + kh) * 7 + 2 * ox + kw;
                return sm4[EN_C3 / 4 + sp * EN_P3 + (i & 3) * 4 + q];
            };
            float4 aq[PD];
#pragma unroll
            for (int p = 0; p < PD; ++p) aq[p] = wfrag(wr, ln, (size_t)(p * 4 + w) * 64);
            float4 bv = bfrag(0);
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                const float4 av = aq[i % PD];
                if (i + PD < 36) aq[i % PD] = wfrag(wr, ln, (size_t)((i + PD) * 4 + w) * 64);
                const float4 bn = bfrag(i + 1 < 36 ? i + 1 : i);
                __builtin_amdgcn_sched_barrier(0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, c1, 0, 0, 0);
                bv = bn;
            }
            if (pv) {
                const int c4 = w * 4 + q;
                const float4 bb = sm4[EN_B4 / 4 + c4];
                float4 v;
                v.x = fmaxf(c0[0] + c1[0] + bb.x, 0.f); v.y = fmaxf(c0[1] + c1[1] + bb.y, 0.f);
                v.z = fmaxf(c0[2] + c1[2] + bb.z, 0.f); v.w = fmaxf(c0[3] + c1[3] + bb.w, 0.f);
                reinterpret_cast<float4*>(a.out + (size_t)img * 576 + px * 64)[c4] = v;
            }
        }
        __syncthreads();        // conv3's output (aliasing the input image buffer) is consumed: the next image may be staged
    }
}

void launch_enc_trunk(const EncArgs& a, hipStream_t st) {
    const int slots = 256 * EFE_ENC_WAVES;
    const int grid = a.rows < slots ? a.rows : slots;
    hipLaunchKernelGGL(k_enc_trunk, dim3(grid), dim3(256), EN_END * sizeof(float), st, a);
}

}  // namespace efe
