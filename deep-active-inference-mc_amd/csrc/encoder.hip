// Fused encoder trunk for gfx950: the four Conv2d(k3, stride 2, no padding) + ReLU layers of
// /root/reference/src/torchmodel.py:85-92 (qs_net.0..7), 64x64x1 -> 31x31x32 -> 15x15x32 -> 7x7x64 -> 3x3x64,
// one image per workgroup, every intermediate in LDS; only the 576-vector (NHWC p*64+c order, the column order the
// packed first dense layer expects) leaves the chip.
//
//   conv1 (Cin = 1) is never materialised: while the conv2 MFMAs run, the VALU recomputes the conv1 activations a
//   lane needs for its next B fragment straight from the 16 KiB input image in LDS (9 FMAs per value, conv1 weights
//   as broadcast LDS reads).  conv2..4 are MFMA tap contractions (v_mfma_f32_32x32x2_f32) over swizzled LDS images.
#include "mfma_pipe.h"

#ifndef EFE_ENC_WAVES
#define EFE_ENC_WAVES 3      // waves per SIMD = workgroups per CU (47 KiB LDS, <= 168 VGPRs)
#endif

namespace efe {

constexpr int EN_IMG = 0;                      // [64][64] input image                       (floats)
constexpr int EN_C2 = 4096;                    // [225 px][32 ch] conv2 output, 8 quads/pixel, quad ^= px & 7
constexpr int EN_C3 = EN_IMG;                  // [49 px][64 ch] conv3 output, 16 quads/pixel, quad ^= px & 15: aliases the input image
                                               // (dead once conv2 is done); 3136 <= 4096 floats
constexpr int EN_W1 = EN_C2 + 225 * 32;        // conv1 weights [9 taps][32 ch] + bias [32]
constexpr int EN_B2 = EN_W1 + 320;             // conv2 / conv3 / conv4 biases [32] [64] [64]: read in the epilogues from LDS (a global
constexpr int EN_B3 = EN_B2 + 32;              // load there exposes an L2 round trip per phase and image)
constexpr int EN_B4 = EN_B3 + 64;
constexpr int EN_END = EN_B4 + 64;             // 11776 floats = 47104 B: three workgroups per CU

__global__ void __launch_bounds__(256, EFE_ENC_WAVES) k_enc_trunk(const EncArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    float4* sm4 = reinterpret_cast<float4*>(smf);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    // conv1 weights + bias live in LDS ([tap][32 ch] so a lane's 4 channels of a chunk are one broadcast ds_read_b128)
    for (int i = tid; i < 320; i += 256) smf[EN_W1 + i] = (i < 288) ? a.w1[i] : a.b1[i - 288];
    if (tid < 160) smf[EN_B2 + tid] = tid < 32 ? a.b2[tid] : tid < 96 ? a.b3[tid - 32] : a.b4[tid - 96];
    const float4* w1s = reinterpret_cast<const float4*>(smf + EN_W1);
    const float4* W2 = reinterpret_cast<const float4*>(a.w2) + lane;      // [9][1][4][64]
    const float4* W3 = reinterpret_cast<const float4*>(a.w3);             // [9][2][4][64] (uniform base, see TapPipe)
    const float4* W4 = reinterpret_cast<const float4*>(a.w4);             // [9][4][4][64], packed for the 16x16x4 form

    for (int img = blockIdx.x; img < a.rows; img += gridDim.x) {
        if (!row_live(a.live, img)) continue;            // a dead row of the call (efe_set_row_mask): workgroup-uniform
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

        // ================= conv2 (with conv1 computed on the fly): 225 output pixels = 8 tiles of 32 ==================
        {
            int ibase[2]; bool pv[2]; int m[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                m[nt] = 32 * (w + 4 * nt) + j;
                pv[nt] = m[nt] < 225;
                const int mm = pv[nt] ? m[nt] : 0;
                const int oy = mm / 15, ox = mm - oy * 15;
                ibase[nt] = (4 * oy) * 64 + 4 * ox;            // image offset of conv1 position (2oy, 2ox)
            }
            f32x16 acc[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
            // B fragment generator: conv1 + ReLU of channels 8*kc + 4*h .. +3 at this lane's conv1 position of tap t
            auto load_patch = [&](int t, float (&patch)[2][9]) {
                const int kh = t / 3, kw = t - kh * 3;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float* ip = smf + EN_IMG + ibase[nt] + (2 * kh) * 64 + 2 * kw;   // conv1 position (2oy+kh, 2ox+kw)
#pragma unroll
                    for (int aa = 0; aa < 3; ++aa)
#pragma unroll
                        for (int bb = 0; bb < 3; ++bb) patch[nt][aa * 3 + bb] = ip[aa * 64 + bb];
                }
            };
            // (packed v_pk_fma_f32 for the two channel pairs was measured: 6 % slower than scalar FMAs here)
            auto gen_b = [&](const float (&patch)[2][9], int kc, int hq, float4 (&bv)[2]) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float4 x = w1s[72 + kc * 2 + hq];                         // bias
#pragma unroll
                    for (int ab = 0; ab < 9; ++ab) {
                        const float4 ww = w1s[ab * 8 + kc * 2 + hq];
                        const float p_ = patch[nt][ab];
                        x.x = fmaf(p_, ww.x, x.x); x.y = fmaf(p_, ww.y, x.y); x.z = fmaf(p_, ww.z, x.z); x.w = fmaf(p_, ww.w, x.w);
                    }
                    bv[nt] = make_float4(fmaxf(x.x, 0.f), fmaxf(x.y, 0.f), fmaxf(x.z, 0.f), fmaxf(x.w, 0.f));
                }
            };
            // software pipeline over the 36 (tap, chunk) steps: while the 8 MFMAs of step i issue, the VALU produces the B
            // fragments of step i+1 and the A fragment of step i+1 is in flight
            float patch[2][9];
            float4 bcur[2], bnxt[2];
            float4 av = W2[0];
            load_patch(0, patch);
            {
                int hq = h; asm volatile("" : "+v"(hq));
                gen_b(patch, 0, hq, bcur);
            }
#pragma unroll 1
            for (int i = 0; i < 36; ++i) {
                const int ni = (i + 1 < 36) ? i + 1 : 35;
                const float4 an = W2[(size_t)ni * 64];
                int hq = h; asm volatile("" : "+v"(hq));        // laundered per step: stops hipcc hoisting the weight reads into 160 registers
                if ((ni & 3) == 0) load_patch(ni >> 2, patch);
                gen_b(patch, ni & 3, hq, bnxt);
                MFMA4(acc[0], av, bcur[0]) MFMA4(acc[1], av, bcur[1])
                av = an; bcur[0] = bnxt[0]; bcur[1] = bnxt[1];
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
                    sm4[EN_C2 / 4 + m[nt] * 8 + (c4 ^ (m[nt] & 7))] = v;
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
                bs[0] = sp * 8; sw[0] = sp & 7;
            }, PackedWIdx{2, 4, mt});
            if (pv) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c4 = mt * 8 + 2 * g4 + h;
                    const float4 bb = sm4[EN_B3 / 4 + c4];
                    float4 v;
                    v.x = fmaxf(acc[0][0][4 * g4 + 0] + bb.x, 0.f); v.y = fmaxf(acc[0][0][4 * g4 + 1] + bb.y, 0.f);
                    v.z = fmaxf(acc[0][0][4 * g4 + 2] + bb.z, 0.f); v.w = fmaxf(acc[0][0][4 * g4 + 3] + bb.w, 0.f);
                    sm4[EN_C3 / 4 + m * 16 + (c4 ^ (m & 15))] = v;
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
                return sm4[EN_C3 / 4 + sp * 16 + (((i & 3) * 4 + q) ^ (sp & 15))];
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
