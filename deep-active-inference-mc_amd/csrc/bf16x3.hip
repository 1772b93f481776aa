// OPT-IN EXPERIMENT, never the default and never the headline: Linear(256, 16384) + ReLU + Dropout(0.5) of the decoder
// (/root/reference/src/torchmodel.py:116-118, `k_fc4` in decoder.hip) on the bf16 matrix pipe with BOTH operands split into three bf16
// planes (engine option "mfma_bf16x3"):
//
//     x = x_hi + x_mid + x_lo   (each a bf16, residuals formed exactly in fp32: 24 mantissa bits = 3 x 8)
//     w . x  ~=  w_lo x_hi + w_hi x_lo + w_mid x_mid + w_mid x_hi + w_hi x_mid + w_hi x_hi        (terms below 2^-24 dropped)
//
// six v_mfma_f32_32x32x16_bf16 with fp32 accumulation per 16 channels instead of eight v_mfma_f32_32x32x2_f32: 6 x 32 cycles against
// 8 x 64, i.e. up to 2.67 x the fp32 MFMA rate at fp32-GEMM accuracy (tools/bf16_split_check.py: max error 1.25e-6 on a K = 576
// contraction, a plain fp32 GEMM has 1.85e-6).  The inputs ARE narrower than the reference's fp32 operands, which is why this is an
// experiment: bench.py reports it under extras.rollout_bf16x3 with its roofline against (bf16 dense peak / 6), every golden fixture is
// run through it at the unchanged tolerances (tests/test_gpu_parity.py::test_bf16x3_*), observed maxima in profiles/r5_observed_errors.txt.
//
// One workgroup = 8 waves (2 per SIMD) x 64 batch rows.  The row tile is split into its three planes while it is staged:
// LDS [64 rows][3 planes][256 k] bf16, 16 bytes of padding per row (1552 B: 16 consecutive rows cover the 64 banks with their
// ds_read_b128).  A wave owns 64 features x 64 rows per step (2 x 2 tiles of 32 x 32), the weights come as pre-split, fragment-major
// planes [32-feature tile][16-channel step][plane][64 lanes][8 bf16] straight from L2 (1 KiB per wave-level load).  Fragment k order:
// lane (x, g) holds channels 16 ks + 8 g .. + 7 of row / column x in BOTH operands (any common bijection contracts the same 16 channels).
#include "mfma_pipe.h"

namespace efe {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int B3_ROWB = 3 * 512 + 16;                 // bytes per staged row: three planes of 256 bf16 + padding
constexpr size_t B3_LDS = (size_t)64 * B3_ROWB;       // 99 328 B: one workgroup per CU

// round-to-nearest-even fp32 -> bf16 (finite inputs), as the upper 16 bits
__host__ __device__ inline uint32_t bf16_rne(float x) {
    uint32_t u = __builtin_bit_cast(uint32_t, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
// x -> (hi, mid, lo) bf16 bit patterns with x = hi + mid + lo up to 2^-24 |x|
__host__ __device__ inline void split3(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = bf16_rne(x);
    const float r1 = x - __builtin_bit_cast(float, hi << 16);
    mid = bf16_rne(r1);
    const float r2 = r1 - __builtin_bit_cast(float, mid << 16);
    lo = bf16_rne(r2);
}

__host__ __device__ inline int fc4b3_spg(int mtiles) { return ((mtiles + 15) / 16 + 7) / 8; }     // steps of 16 feature tiles, dealt to 8 groups

__global__ void __launch_bounds__(512, 1) k_fc4_b3(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);           // 0 .. 7
    const int j = lane & 31, g = lane >> 5;
    // persistent, XCD-aware and balanced exactly like k_fc4: workgroup b serves feature group b & 7, an equal contiguous range of that
    // group's (64-row tile, 512-feature step) pairs
    const int fgrp = blockIdx.x & 7;
    const int nper = gridDim.x >> 3, k = blockIdx.x >> 3;
    const int SPG = fc4b3_spg(a.mtiles);
    const int nsteps = ((a.n_pix + 63) / 64) * SPG;
    const int q0 = (int)(((long)nsteps * k) / nper), q1 = (int)(((long)nsteps * (k + 1)) / nper);
    const __amdgpu_buffer_rsrc_t wr = wrsrc(a.Wb3);
    const unsigned ln = (unsigned)lane * 16u;
    // this lane's B fragments: row nt * 32 + j, plane p, step ks at byte  row * B3_ROWB + p * 512 + ks * 32 + g * 16
    const unsigned char* brow[2] = {smb + (size_t)j * B3_ROWB + g * 16, smb + (size_t)(32 + j) * B3_ROWB + g * 16};
    int cur_rt = -1;
    uint32_t krow[2] = {0, 0}, kstream[2] = {0, 0}, kstage[2] = {0, 0};
    bool rv[2] = {false, false};
#pragma unroll 1
    for (int q = q0; q < q1; ++q) {
        const int rt = q / SPG, fs = q - rt * SPG;
        const int row0 = rt * 64;
        if (rt != cur_rt) {
            if (cur_rt >= 0) __syncthreads();
            cur_rt = rt;
            const f32x4* X = reinterpret_cast<const f32x4*>(a.X);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = it * 512 + tid;                        // 64 rows x 64 quads
                const int r = idx >> 6, c4 = idx & 63;
                const int gr = row0 + r;
                const f32x4 v = (gr < a.n_pix) ? X[(size_t)gr * 64 + c4] : (f32x4)(0.f);
                uint32_t hi[4], mid[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split3(v[e], hi[e], mid[e], lo[e]);
                unsigned char* d = smb + (size_t)r * B3_ROWB + c4 * 8;
                *reinterpret_cast<uint2*>(d) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
                *reinterpret_cast<uint2*>(d + 512) = make_uint2(mid[0] | (mid[1] << 16), mid[2] | (mid[3] << 16));
                *reinterpret_cast<uint2*>(d + 1024) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
            }
            __syncthreads();
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {                            // dropout keys of this lane's two rows
                const int m = row0 + nt * 32 + j;
                rv[nt] = m < a.n_pix;
                const int mg = a.m0 + (rv[nt] ? m : 0);
                const int gq = mg / a.rows_per_group;
                krow[nt] = global_row(a.gm.ids, a.gm.ids_div, mg - gq * a.rows_per_group, a.row_offset);
                const uint2 key = group_key(a.gm, gq);
                kstream[nt] = key.x; kstage[nt] = key.y;
            }
        }
        const int mt0 = (fgrp * SPG + fs) * 16 + 2 * w;                 // this wave's first 32-feature tile
        if (mt0 >= a.mtiles) continue;                                  // wave-uniform (mtiles is even)
        f32x16 acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        float4 bq[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) bq[mt][g4] = *reinterpret_cast<const float4*>(a.bias + (mt0 + mt) * 32 + 8 * g4 + 4 * g);
        // fragment of (tile mt, step ks, plane p): float4 index ((mt * 16 + ks) * 3 + p) * 64 + lane
        auto afrag = [&](int mt, int ks, int p) -> float4 { return wfrag(wr, ln, (size_t)(((mt0 + mt) * 16 + ks) * 3 + p) * 64); };
        auto bfrag = [&](int nt, int ks, int p) -> float4 { return *reinterpret_cast<const float4*>(brow[nt] + p * 512 + ks * 32); };
        // fragments of step ks live in buffer set ks & 1; the loop is fully unrolled so that every index is static (hipcc copies a
        // software-pipeline buffer it cannot rename: 48 v_mov per step and an s_waitcnt vmcnt(0) on the loads that were meant to stay in flight)
        float4 af[2][2][3], bf[2][2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) { af[0][t][p] = afrag(t, 0, p); bf[0][t][p] = bfrag(t, 0, p); }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int cb = ks & 1, nb = cb ^ 1;
            if (ks < 15) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int p = 0; p < 3; ++p) { af[nb][t][p] = afrag(t, ks + 1, p); bf[nb][t][p] = bfrag(t, ks + 1, p); }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the six products, smallest first: (lo, hi) (hi, lo) (mid, mid) (mid, hi) (hi, mid) (hi, hi); planes 0 = hi, 1 = mid, 2 = lo
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[cb][mt][PA[pr]]), __builtin_bit_cast(bf16x8, bf[cb][nt][PB[pr]]),
                                                                              acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: the same as k_fc4 (bias, ReLU, dropout mask from Philox, NHWC store)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (!rv[nt]) continue;
            const uint4 rnd = noise_words(a.k0, a.k1, a.tag, (uint32_t)((mt0 * 32) >> 7), krow[nt], kstream[nt], kstage[nt]);
            float* yp = a.Y + (size_t)(row0 + nt * 32 + j) * a.ldy;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int co = (mt0 + mt) * 32 + 8 * g4 + 4 * g;
                    const float4 bb = bq[mt][g4];
                    const uint32_t word = ((co >> 5) & 3) == 0 ? rnd.x : ((co >> 5) & 3) == 1 ? rnd.y : ((co >> 5) & 3) == 2 ? rnd.z : rnd.w;
                    float v[4] = {acc[mt][nt][4 * g4 + 0] + bb.x, acc[mt][nt][4 * g4 + 1] + bb.y,
                                  acc[mt][nt][4 * g4 + 2] + bb.z, acc[mt][nt][4 * g4 + 3] + bb.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ((word >> ((co + e) & 31)) & 1u) ? fmaxf(v[e], 0.f) * 2.0f : 0.0f;
                    *reinterpret_cast<float4*>(yp + co) = make_float4(v[0], v[1], v[2], v[3]);
                }
        }
    }
}

int init_bf16x3_kernels() {
    return hipFuncSetAttribute((const void*)k_fc4_b3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3_LDS) != hipSuccess;
}

void launch_fc4_b3(const GemmArgs& a, hipStream_t st) {
    // persistent: one 8-wave workgroup per CU, 8 feature groups x 32 workgroups, each with >= 1 (row tile, step) pair
    const int nsteps = ((a.n_pix + 63) / 64) * fc4b3_spg(a.mtiles);
    const int nper = nsteps < 32 ? nsteps : 32;
    hipLaunchKernelGGL(k_fc4_b3, dim3(8 * nper), dim3(512), B3_LDS, st, a);
}

// host side of efe_commit_weights: W [out][in = 256] (rows already in the engine's NHWC feature order through row_perm) -> three bf16
// planes, fragment-major [out / 32][16][3][64 lanes][8]
void pack_bf16x3(const float* W, const int* row_perm, int out, int in, uint16_t* dst) {
    const int mtiles = out / 32;
    for (int mt = 0; mt < mtiles; ++mt)
        for (int ks = 0; ks < in / 16; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 8; ++s) {
                    const int co = mt * 32 + (lane & 31), ci = ks * 16 + 8 * (lane >> 5) + s;
                    uint32_t p[3];
                    split3(W[(size_t)(row_perm ? row_perm[co] : co) * in + ci], p[0], p[1], p[2]);
                    for (int pl = 0; pl < 3; ++pl) dst[((((size_t)mt * (in / 16) + ks) * 3 + pl) * 64 + lane) * 8 + s] = (uint16_t)p[pl];
                }
}

}  // namespace efe
