// OPT-IN EXPERIMENTS, never the default and never the headline: the four big contractions of the decoder -- Linear(256, 16384) + ReLU +
// Dropout(0.5), ConvTranspose2d(64, 64, s1), ConvTranspose2d(64, 64, s2), ConvTranspose2d(64, 32, s2) (/root/reference/src/torchmodel.py:116-125;
// `k_fc4`, `k_dec_a`, `k_dec_b4` in decoder.hip are the exact-fp32 kernels) -- on the 16-bit matrix pipe with BOTH operands split into
// several 16-bit planes and fp32 accumulation.  Two schemes, one source (structs SchB3 / SchH2 below; engine options):
//
//   "mfma_bf16x3"   x = x_hi + x_mid + x_lo  (three bf16, residuals formed exactly in fp32: 24 mantissa bits = 3 x 8)
//                   w . x ~= w_lo x_hi + w_hi x_lo + w_mid x_mid + w_mid x_hi + w_hi x_mid + w_hi x_hi          (terms below 2^-24 dropped)
//                   six v_mfma_f32_32x32x16_bf16 per 16 channels instead of eight v_mfma_f32_32x32x2_f32: 6 x 32 cycles against 8 x 64
//   "mfma_f16x2"    x = x_hi + x_lo  (two fp16: 22 mantissa bits; the weights scaled by a power of two so that their low planes are normal)
//                   w . x ~= w_lo x_hi + w_hi x_lo + w_hi x_hi                                                   (the term below 2^-22 dropped)
//                   three v_mfma_f32_32x32x16_f16 per 16 channels: 3 x 32 cycles, two thirds of the operand bytes
//
// Operand-representation error on a K = 576 decoder-like contraction (tools/f16_split_check.py): 8.6e-8 / 1.0e-6, a plain fp32 GEMM's accumulation
// 5.0e-6.  The inputs ARE narrower than the reference's fp32 operands, which is why these are experiments: bench.py reports them under
// extras.rollout_bf16x3 / extras.rollout_f16x2 with their rooflines against (16-bit dense peak / products), every golden fixture and the
// distribution tests run through both at the unchanged tolerances (tests/test_gpu_parity.py::test_split_operands_*,
// tests/test_noise_statistics.py::test_device_noise_distributions_with_split_operands), observed maxima in profiles/r6_observed_errors.txt.
//
// k_fc4_b3: one workgroup = 8 waves (2 per SIMD) x 64 batch rows.  The row tile is split into its planes while it is staged:
// LDS [64 rows][planes][256 k] x 16 bit, 16 bytes of padding per row (1552 / 1040 B: 16 consecutive rows cover the 64 banks with their
// ds_read_b128).  A wave owns 64 features x 64 rows per step (2 x 2 tiles of 32 x 32), the weights come as pre-split, fragment-major
// planes [32-feature tile][16-channel step][plane][64 lanes][8 x 16 bit] straight from L2 (1 KiB per wave-level load).  Fragment k order:
// lane (x, g) holds channels 16 ks + 8 g .. + 7 of row / column x in BOTH operands (any common bijection contracts the same 16 channels).
#include <cmath>
#include "mfma_pipe.h"

namespace efe {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// LDS-DMA of 1 KiB per wave: lane l copies 16 bytes from its own global address to LDS byte address lds_dst + 16 l (lds_dst wave-uniform).
// As inline asm, not __builtin_amdgcn_global_load_lds: hipcc orders every later ds_read behind the builtin with s_waitcnt vmcnt(0) (it
// cannot see that the copy targets the OTHER weight buffer), which exposes the copy's whole latency; the asm form is invisible to its
// counters, so the caller drains it itself (glds_drain) in front of the barrier that publishes the buffer.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int NPL> struct Fc4L {
    static constexpr int ROWB = NPL * 512 + 16;                   // bytes per staged row: NPL planes of 256 x 16 bit + padding
    // 32-row tiles per workgroup pass: 64 rows with three planes (99 328 B), 96 rows with two (99 840 B) -- a weight fragment crosses the L1
    // once per wave and (row tile, feature step): 1.5 x the rows = two thirds of the fragment traffic, which is what bounds this kernel
    static constexpr int NT = NPL == 2 ? 3 : 2;        // (four tiles need 128 accumulator + 96 fragment registers: spills)
    static constexpr size_t LDS = (size_t)32 * NT * ROWB;
};

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

// the same split for a PAIR of values on the device: v_cvt_pk_bf16_f32 (round to nearest even in hardware, two values per instruction) ->
// packed (hi, mid, lo) words, element 0 in the low half; 9 VALU instructions per pair against ~40 for two integer split3
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pk(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    f32x2 v; v.x = x0; v.y = x1;
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
    f32x2 hf; hf.x = __builtin_bit_cast(float, hi << 16); hf.y = __builtin_bit_cast(float, hi & 0xffff0000u);
    const f32x2 r1 = v - hf;
    mid = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2_t));
    f32x2 mf; mf.x = __builtin_bit_cast(float, mid << 16); mf.y = __builtin_bit_cast(float, mid & 0xffff0000u);
    const f32x2 r2 = r1 - mf;
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf16x2_t));
}

// ---------------------------------------------------------------------------------------------------------
// The two operand splits the kernels below are instantiated for (engine options "mfma_bf16x3" / "mfma_f16x2"):
//   SchB3: x = hi + mid + lo as three bf16 (3 x 8 mantissa bits), six products; error of an fp32 GEMM, every fixture at unchanged tolerances.
//   SchH2: x = hi + lo as two fp16 (2 x 11 bits), three products  w_lo x_hi + w_hi x_lo + w_hi x_hi  on v_mfma_f32_32x32x16_f16 -- half
//          the matrix work, two thirds of the operand bytes.  fp16 has 5 exponent bits: the WEIGHTS are scaled by a power of two at pack
//          time (largest |w| just below 2^14; the kernels multiply the accumulators back, exactly) so that their low planes are normal
//          numbers; activations stay unscaled -- the matrix pipe and v_cvt_pk_f16_f32 keep fp16 denormals (tools/ubench/f16_probe.hip), so
//          an activation below 2^-3 loses at most 2^-25 absolutely -- and must stay below 65 504: a larger one is DETECTED where it is
//          split and the row / image poisoned (SchH2::overflow below): loud, never silently wrong.  Operand representation error 2^-22 relative: tools/f16_split_check.py.
// Plane 0 = hi.  PA / PB: the weight / activation plane of product pr, small terms first.
// ---------------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
struct SchB3 {
    static constexpr int NPL = 3, NPR = 6, MODE = 1;
    static constexpr bool SCALED = false;
    static constexpr int PA(int pr) { constexpr int t[6] = {2, 0, 1, 1, 0, 0}; return t[pr]; }
    static constexpr int PB(int pr) { constexpr int t[6] = {0, 2, 1, 0, 1, 0}; return t[pr]; }
    static __device__ __forceinline__ f32x16 mfma(const float4& a, const float4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void split_pk(float x0, float x1, uint32_t (&pl)[3]) { split3_pk(x0, x1, pl[0], pl[1], pl[2]); }
    static __device__ __forceinline__ uint32_t overflow(uint32_t) { return 0u; }       // bf16 has fp32's exponent range
};
struct SchH2 {
    static constexpr int NPL = 2, NPR = 3, MODE = 2;
    static constexpr bool SCALED = true;
    static constexpr int PA(int pr) { constexpr int t[3] = {1, 0, 0}; return t[pr]; }
    static constexpr int PB(int pr) { constexpr int t[3] = {0, 1, 0}; return t[pr]; }
    static __device__ __forceinline__ f32x16 mfma(const float4& a, const float4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void split_pk(float x0, float x1, uint32_t (&pl)[2]) {
        f32x2 v; v.x = x0; v.y = x1;
        const f16x2_t hh = __builtin_convertvector(v, f16x2_t);           // v_cvt_pk_f16_f32: round to nearest even, denormals kept
        pl[0] = __builtin_bit_cast(uint32_t, hh);
        const f32x2 r = v - __builtin_convertvector(hh, f32x2);          // exact
        pl[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2_t));
    }
    // non-zero iff one of the two NON-NEGATIVE values behind the packed high plane `hi` did not fit fp16 (exponent field 31: inf, or a NaN
    // that came in) -- every activation these kernels split is the output of a ReLU.  The kernels OR it over what they stage and POISON the
    // outputs of an image / row that saw it (+inf downstream, NaN in the per-image sums): the integer-max ReLU would otherwise turn the
    // (negative-signed) NaN of inf - inf into a finite, silently wrong number.
    static __device__ __forceinline__ uint32_t overflow(uint32_t hi) { return (hi + 0x04000400u) & 0x80008000u; }
};
// host side: x (already scaled) -> NPL 16-bit planes
inline void split_host(int mode, float x, uint32_t (&p)[3]) {
    if (mode == 1) { split3(x, p[0], p[1], p[2]); return; }
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    p[0] = __builtin_bit_cast(uint16_t, hi); p[1] = __builtin_bit_cast(uint16_t, lo); p[2] = 0;
}
// the power of two the weights of a layer are multiplied by before the fp16 split: the largest magnitude lands in [2^13, 2^14)
float split_weight_scale(int mode, const float* w, size_t n) {
    if (mode != 2) return 1.0f;
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = fmaxf(mx, fabsf(w[i]));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.0f;
    int e; frexpf(mx, &e);                        // mx = f * 2^e, f in [0.5, 1)
    return ldexpf(1.0f, 14 - e);
}

__host__ __device__ inline int fc4b3_spg(int mtiles) { return ((mtiles + 15) / 16 + 7) / 8; }     // steps of 16 feature tiles, dealt to 8 groups

template <class SC>
__global__ void __launch_bounds__(512, 1) k_fc4_b3(const GemmArgs a) {
    constexpr int NPL = SC::NPL, B3_ROWB = Fc4L<SC::NPL>::ROWB, NT = Fc4L<SC::NPL>::NT, RT = 32 * NT;      // NT 32-row tiles per workgroup pass
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
    const int nsteps = ((a.n_pix + RT - 1) / RT) * SPG;
    const int q0 = (int)(((long)nsteps * k) / nper), q1 = (int)(((long)nsteps * (k + 1)) / nper);
    const __amdgpu_buffer_rsrc_t wr = wrsrc(a.Wb3);
    const unsigned ln = (unsigned)lane * 16u;
    // this lane's B fragments: row nt * 32 + j, plane p, step ks at byte  row * B3_ROWB + p * 512 + ks * 32 + g * 16
    const unsigned char* brow[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) brow[nt] = smb + (size_t)(32 * nt + j) * B3_ROWB + g * 16;
    int cur_rt = -1, tpar = 1;
    // fp16 split: rows whose input did not fit fp16, per row tile in flight (two arrays: tile k uses k & 1, the other one is cleared meanwhile)
    __shared__ int sflag[2][32 * NT];
    if (tid < 32 * NT) { sflag[0][tid] = 0; sflag[1][tid] = 0; }
    __syncthreads();
    uint32_t krow[NT] = {}, kstream[NT] = {}, kstage[NT] = {};
    bool rv[NT] = {};
#pragma unroll 1
    for (int q = q0; q < q1; ++q) {
        const int rt = q / SPG, fs = q - rt * SPG;
        const int row0 = rt * RT;
        if (rt != cur_rt) {
            if (cur_rt >= 0) __syncthreads();
            cur_rt = rt;
            tpar ^= 1;
            const f32x4* X = reinterpret_cast<const f32x4*>(a.X);
#pragma unroll
            for (int it = 0; it < RT / 8; ++it) {
                const int idx = it * 512 + tid;                        // RT rows x 64 quads
                const int r = idx >> 6, c4 = idx & 63;
                const int gr = row0 + r;
                const f32x4 v = (gr < a.n_pix) ? X[(size_t)gr * 64 + c4] : (f32x4)(0.f);
                uint32_t pa_[NPL], pb_[NPL];
                SC::split_pk(v[0], v[1], pa_); SC::split_pk(v[2], v[3], pb_);
                if (SC::SCALED && (SC::overflow(pa_[0]) | SC::overflow(pb_[0]))) sflag[tpar][r] = 1;        // (rare; every writer stores the same value)
                unsigned char* d = smb + (size_t)r * B3_ROWB + c4 * 8;
#pragma unroll
                for (int p = 0; p < NPL; ++p) *reinterpret_cast<uint2*>(d + p * 512) = make_uint2(pa_[p], pb_[p]);
            }
            __syncthreads();
            if (SC::SCALED && tid < 32 * NT) sflag[tpar ^ 1][tid] = 0;  // (the previous tile's flags: its epilogues ended in front of this tile's first barrier)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {                           // dropout keys of this lane's rows
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
        f32x16 acc[2][NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        const float ws_inv = SC::SCALED ? a.wb3_scale_inv : 1.0f;        // fp16 split: the accumulators hold (weight scale) x W x
        // fragment of (tile mt, step ks, plane p): float4 index ((mt * 16 + ks) * 3 + p) * 64 + lane
        auto afrag = [&](int mt, int ks, int p) -> float4 { return wfrag(wr, ln, (size_t)(((mt0 + mt) * 16 + ks) * NPL + p) * 64); };
        auto bfrag = [&](int nt, int ks, int p) -> float4 { return *reinterpret_cast<const float4*>(brow[nt] + p * 512 + ks * 32); };
        // fragments of step ks live in buffer set ks & 1; the loop is fully unrolled so that every index is static (hipcc copies a
        // software-pipeline buffer it cannot rename: 48 v_mov per step and an s_waitcnt vmcnt(0) on the loads that were meant to stay in flight)
        float4 af[2][2][NPL], bf[2][NT][NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
            for (int t = 0; t < 2; ++t) af[0][t][p] = afrag(t, 0, p);
#pragma unroll
            for (int t = 0; t < NT; ++t) bf[0][t][p] = bfrag(t, 0, p);
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int cb = ks & 1, nb = cb ^ 1;
            if (ks < 15) {
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) af[nb][t][p] = afrag(t, ks + 1, p);
#pragma unroll
                    for (int t = 0; t < NT; ++t) bf[nb][t][p] = bfrag(t, ks + 1, p);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the products, smallest first (SchB3: (lo, hi) (hi, lo) (mid, mid) (mid, hi) (hi, mid) (hi, hi); SchH2: (lo, hi) (hi, lo) (hi, hi))
#pragma unroll
            for (int pr = 0; pr < SC::NPR; ++pr)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = SC::mfma(af[cb][mt][SC::PA(pr)], bf[cb][nt][SC::PB(pr)], acc[mt][nt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: the same as k_fc4 (bias, ReLU, dropout mask from Philox, NHWC store); the bias is requested here, not held through the
        // contraction (32 registers that the three-tile form does not have)
        float4 bq[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) bq[mt][g4] = *reinterpret_cast<const float4*>(a.bias + (mt0 + mt) * 32 + 8 * g4 + 4 * g);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (!rv[nt]) continue;
            const uint4 rnd = noise_words(a.k0, a.k1, a.tag, (uint32_t)((mt0 * 32) >> 7), krow[nt], kstream[nt], kstage[nt]);
            float* yp = a.Y + (size_t)(row0 + nt * 32 + j) * a.ldy;
            const bool rbad = SC::SCALED && sflag[tpar][nt * 32 + j] != 0;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int co = (mt0 + mt) * 32 + 8 * g4 + 4 * g;
                    const float4 bb = bq[mt][g4];
                    const uint32_t word = ((co >> 5) & 3) == 0 ? rnd.x : ((co >> 5) & 3) == 1 ? rnd.y : ((co >> 5) & 3) == 2 ? rnd.z : rnd.w;
                    float v[4];
                    if (SC::SCALED) {
                        v[0] = __builtin_fmaf(acc[mt][nt][4 * g4 + 0], ws_inv, bb.x); v[1] = __builtin_fmaf(acc[mt][nt][4 * g4 + 1], ws_inv, bb.y);
                        v[2] = __builtin_fmaf(acc[mt][nt][4 * g4 + 2], ws_inv, bb.z); v[3] = __builtin_fmaf(acc[mt][nt][4 * g4 + 3], ws_inv, bb.w);
                    } else {
                        v[0] = acc[mt][nt][4 * g4 + 0] + bb.x; v[1] = acc[mt][nt][4 * g4 + 1] + bb.y;
                        v[2] = acc[mt][nt][4 * g4 + 2] + bb.z; v[3] = acc[mt][nt][4 * g4 + 3] + bb.w;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ((word >> ((co + e) & 31)) & 1u) ? fmaxf(v[e], 0.f) * 2.0f : 0.0f;
                    if (SC::SCALED && rbad) v[0] = v[1] = v[2] = v[3] = __builtin_inff();      // an input of this row did not fit fp16: the row is poisoned (k_dec_a_b3 passes it on)
                    *reinterpret_cast<float4*>(yp + co) = make_float4(v[0], v[1], v[2], v[3]);
                }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// k_dec_a_b3: the two fused layers of k_dec_a (decoder.hip: ConvTranspose2d(64, 64, s1) + ReLU -> ConvTranspose2d(64, 64, s2) + ReLU,
// /root/reference/src/torchmodel.py:120-123) on the bf16 pipe, same option.  EVERY operand comes from LDS -- the ablation of k_fc4_b3
// (profiles/r5_fc4_b3_ablation.txt) shows the three-plane form bound by the return path of its global fragment loads (~30 B/clk and CU):
//   * the image as three bf16 planes, [257 pixels][3 planes][64 channels] + 16 B of padding per pixel (400 B: 16 consecutive pixels of
//     a ds_read_b128 cover the 64 banks); layer 1's output is split and written back IN PLACE as layer 2's input, as in k_dec_a;
//   * the weights of ONE tap (all 64 x 64, three planes, fragment-major = lane-linear: 24 KiB) in a double buffer, staged by the whole
//     workgroup one tap ahead -- every wave reads the same fragments, so they cross the L1 once per workgroup instead of once per wave.
// One workgroup per CU (150 KB of LDS), persistent over images (static stride, the next image prefetched into registers over the last
// four taps).  Shipped form: 8 waves, a wave owns 64 channels x 32 pixels (2 x 1 tiles: 9 ds_read_b128 per 12 MFMAs and 16-channel step);
// the 4-wave 2 x 2 form (12 reads per 24 MFMAs, one wave per SIMD) is 2.5 % slower (tools/ubench/patches/da3_4waves.py).  y2 leaves in
// k_dec_a's fp32 layout: k_dec_b4 is unchanged.
// ---------------------------------------------------------------------------------------------------------
template <int NPL_> struct Da3L {
    static constexpr int PXB = NPL_ * 128 + 16;                 // bytes per staged pixel
    static constexpr int IMG = 257 * PXB;                       // image + zero pixel
    static constexpr int SLAB = 2 * 4 * NPL_ * 1024;            // one tap's weights: [2 mt][4 ks][NPL planes][64 lanes][16 B]
    static constexpr int W0 = (IMG + 255) & ~255;               // byte offset of weight buffer 0
    static constexpr int BIAS = W0 + 2 * SLAB;                  // two bias vectors (fp32)
    static constexpr size_t LDS = BIAS + 2 * 64 * sizeof(float);
};

// NTW = 32-pixel tiles per wave: 2 -> four waves (one per SIMD, 2 x 2 register tiles), 1 -> eight waves (two per SIMD, 2 x 1 tiles: nine
// fragment reads per 12 MFMAs instead of 12 per 24, but a second wave on every SIMD fills the pipe while the first one stages, splits,
// stores or waits at a barrier)
#ifndef EFE_DA3_NTW
#define EFE_DA3_NTW 1
#endif
template <class SC, int NTW>
__global__ void __launch_bounds__(512 / NTW, 1) k_dec_a_b3(const DecAArgs a) {
    constexpr int NPL = SC::NPL, NPR = SC::NPR;
    constexpr int DA3_PXB = Da3L<NPL>::PXB, DA3_SLAB = Da3L<NPL>::SLAB, DA3_W0 = Da3L<NPL>::W0, DA3_BIAS = Da3L<NPL>::BIAS;
    constexpr int NW = 8 / NTW, NTHR = 64 * NW, NPF = 4096 / NTHR, PP = 8 * NPL / NW;  // waves, threads, float4 of an image per thread, DMA pieces per wave and tap
    constexpr int MF = 2 * NPR * NTW, NA = 2 * NPL, NL = NA + NPL * NTW;               // MFMAs and fragment reads (NA of them weights) of one 16-channel step
    extern __shared__ __attribute__((aligned(16))) unsigned char sm3[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, g = lane >> 5;
    if ((int)blockIdx.x >= a.rows) return;
    const float4* bias4 = reinterpret_cast<const float4*>(sm3 + DA3_BIAS);
    if (tid < 16) reinterpret_cast<float4*>(sm3 + DA3_BIAS)[tid] = reinterpret_cast<const float4*>(a.b1)[tid];
    else if (tid < 32) reinterpret_cast<float4*>(sm3 + DA3_BIAS)[tid] = reinterpret_cast<const float4*>(a.b2)[tid - 16];
    for (int i = tid; i < DA3_PXB / 4; i += NTHR) reinterpret_cast<uint32_t*>(sm3 + 256 * DA3_PXB)[i] = 0u;      // the zero pixel
    const float4* W1 = reinterpret_cast<const float4*>(a.w1b3);
    const float4* W2 = reinterpret_cast<const float4*>(a.w2b3);
    // packed tap index kh * 3 + kw of tap t of output parity par of the stride-2 layer (the order of ConvT2Addr, mfma_pipe.h)
    auto l2_wt = [](int par, int t) -> int {
        const int ph = par >> 1, pw = par & 1;
        const int th = t / (1 + pw), tw = t - th * (1 + pw);
        return (ph ? (th ? 2 : 0) : 1) * 3 + (pw ? (tw ? 2 : 0) : 1);
    };
    // a tap's 24 KiB of packed weights (tap index T = 0 .. 17: layer 1's taps, then layer 2's in the order its parities use them) go
    // global -> LDS buffer T & 1 by LDS-DMA: 1 KiB per wave-instruction, destination = a wave-uniform base + lane * 16 -- exactly the
    // fragment-major layout; no staging registers, no ds_write.  Wave w copies pieces PP w .. PP w + PP - 1
    const unsigned lds_w0 = (unsigned)(size_t)(sm3 + DA3_W0) + (unsigned)(PP * w) * 1024u;
    auto slab_dma = [&](const float4* src, int buf) {
        const char* sp = reinterpret_cast<const char*>(src) + (size_t)(PP * w) * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < PP; ++i) glds16(sp + i * 1024, lds_w0 + (unsigned)(buf * DA3_SLAB + i * 1024));
    };
    // fp32 quad (4 consecutive channels c .. c + 3 of one pixel) -> the three planes of that pixel
    uint32_t ovf = 0;                                    // fp16 split: something this thread staged did not fit fp16
    auto put_planes = [&](unsigned char* px, int c, const float v0, const float v1, const float v2, const float v3) {
        uint32_t pa_[NPL], pb_[NPL];
        SC::split_pk(v0, v1, pa_); SC::split_pk(v2, v3, pb_);
        ovf |= SC::overflow(pa_[0]) | SC::overflow(pb_[0]);
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<uint2*>(px + p * 128 + c * 2) = make_uint2(pa_[p], pb_[p]);
    };
    // the image's overflow flag: word (image count & 1), set behind the two staging phases, read by the layer-2 stores (tap barriers in
    // between), the other word cleared meanwhile
    __shared__ int sovf[2];
    if (tid < 2) sovf[tid] = 0;
    int nimgs = 0;
    f32x4 pf[NPF];                                       // the next image: 64 KiB / NTHR threads
    {
        const f32x4* X = reinterpret_cast<const f32x4*>(a.x4) + (size_t)blockIdx.x * 4096;
#pragma unroll
        for (int it = 0; it < NPF; ++it) pf[it] = X[it * NTHR + tid];
    }
    // this wave's 32 NTW pixels: image rows 2 NTW w .. as NTW tiles of two rows each
    const int pcol = j & 15, prow0 = 2 * NTW * w + (j >> 4);
    const unsigned char* abase = sm3 + DA3_W0 + lane * 16;

    for (int img = blockIdx.x; img < a.rows; img += gridDim.x) {
        const bool live = row_live(a.live, img);
        const int nimg = img + (int)gridDim.x;
        const int fpar = nimgs & 1;
        ++nimgs;
        __syncthreads();                                  // every wave is done with the previous image's planes
        if (SC::SCALED && tid == 0) sovf[fpar ^ 1] = 0;
        if (live) {
#pragma unroll
            for (int it = 0; it < NPF; ++it) {
                const int idx = it * NTHR + tid;          // pixel idx >> 4, channel quad idx & 15
                put_planes(sm3 + (size_t)(idx >> 4) * DA3_PXB, 4 * (idx & 15), pf[it][0], pf[it][1], pf[it][2], pf[it][3]);
            }
            if (SC::SCALED && ovf) { sovf[fpar] = 1; ovf = 0; }
        }
        slab_dma(W1, 0);
        f32x16 acc[2][NTW];
        auto acc_init = [&](int boff, float wsc) {        // the accumulators start at the bias (x the fp16 split's weight scale): register e of tile mt holds channel 32 mt + (e & 3) + 8 (e >> 2) + 4 g
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float4 bb = bias4[boff + mt * 8 + 2 * g4 + g];
                    if (SC::SCALED) { bb.x *= wsc; bb.y *= wsc; bb.z *= wsc; bb.w *= wsc; }
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) { acc[mt][nt][4 * g4] = bb.x; acc[mt][nt][4 * g4 + 1] = bb.y; acc[mt][nt][4 * g4 + 2] = bb.z; acc[mt][nt][4 * g4 + 3] = bb.w; }
                }
        };
        // one tap: B views at pixel byte offsets pb[nt] (+ g * 16), weights in buffer T & 1 (copied during the previous tap); the next tap's
        // slab is copied during this one.  Every LDS read and every DMA piece sits between two MFMAs (with one wave per SIMD a burst of 12
        // ds_read_b128 in front of a step's 24 MFMAs cost as much time as the MFMAs: profiles/r5_dec_a_b3_ablation.txt): fragment read l of
        // step ks + 1 follows MFMA l MF / NL of step ks, the DMA pieces follow the middle MFMA of steps 0 .. 2.
        auto tap = [&](int T, const int (&pb)[NTW], const float4* next) {
            glds_drain();                                 // this wave's pieces of slab T have landed ...
            __syncthreads();                              // ... and everybody's; nobody reads buffer (T + 1) & 1 any more
            const unsigned char* ab = abase + (T & 1) * DA3_SLAB;
            const char* dsp = reinterpret_cast<const char*>(next) + (size_t)(PP * w) * 1024 + lane * 16;
            const unsigned ddp = lds_w0 + (unsigned)(((T + 1) & 1) * DA3_SLAB);
            if (T == 14) {                                // the next image, requested over the last four taps (held in registers only from here)
                const f32x4* X = reinterpret_cast<const f32x4*>(a.x4) + (size_t)(nimg < a.rows ? nimg : img) * 4096;
#pragma unroll
                for (int it = 0; it < NPF; ++it) pf[it] = X[it * NTHR + tid];
            }
            float4 af[2][2][NPL], bf[2][NTW][NPL];
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
#pragma unroll
                for (int t = 0; t < 2; ++t) af[0][t][p] = *reinterpret_cast<const float4*>(ab + ((t * 4 + 0) * NPL + p) * 1024);
#pragma unroll
                for (int t = 0; t < NTW; ++t) bf[0][t][p] = *reinterpret_cast<const float4*>(sm3 + pb[t] + p * 128);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cb = ks & 1, nb = cb ^ 1;
#pragma unroll
                for (int m = 0; m < MF; ++m) {
                    const int pr = m / (2 * NTW), mt = (m / NTW) & 1, nt = m % NTW;
                    acc[mt][nt] = SC::mfma(af[cb][mt][SC::PA(pr)], bf[cb][nt][SC::PB(pr)], acc[mt][nt]);
                    if (ks < 3) {
#pragma unroll
                        for (int l = 0; l < NL; ++l) {
                            if (l * MF / NL != m) continue;
                            if (l < NA) af[nb][l / NPL][l % NPL] = *reinterpret_cast<const float4*>(ab + (((l / NPL) * 4 + ks + 1) * NPL + l % NPL) * 1024);
                            else bf[nb][(l - NA) / NPL][(l - NA) % NPL] = *reinterpret_cast<const float4*>(sm3 + pb[(l - NA) / NPL] + ((l - NA) % NPL) * 128 + (ks + 1) * 32);
                        }
                        if (next != nullptr) {                  // DMA piece p of the next slab: behind MFMA MF / 2 + p / 3 of step p % 3
#pragma unroll
                            for (int pp = 0; pp < PP; ++pp)
                                if (pp % 3 == ks && m == MF / 2 + pp / 3) glds16(dsp + pp * 1024, ddp + (unsigned)(pp * 1024));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        // ---------------- layer 1: out[oh, ow] = sum_{kh, kw} in[oh + 1 - kh, ow + 1 - kw] . W[:, :, kh, kw] -------------------------------
        acc_init(0, a.w1s);
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t - kh * 3;
            int pb[NTW];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int sy = prow0 + 2 * nt + 1 - kh, sx = pcol + 1 - kw;
                const bool ok = sy >= 0 && sy < 16 && sx >= 0 && sx < 16;
                pb[nt] = (ok ? sy * 16 + sx : 256) * DA3_PXB + g * 16;
            }
            tap(t, pb, t < 8 ? W1 + (size_t)(t + 1) * (DA3_SLAB / 16) : W2 + (size_t)l2_wt(0, 0) * (DA3_SLAB / 16));
        }
        __syncthreads();                                  // every wave is done reading the input planes
        if (live) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                unsigned char* px = sm3 + (size_t)(32 * NTW * w + 32 * nt + j) * DA3_PXB;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                    {
                        const float u = SC::SCALED ? a.w1s_inv : 1.0f;      // (fp16 split: the weight scale is undone before the activation is split again)
                        put_planes(px, mt * 32 + 8 * g4 + 4 * g, SC::SCALED ? relu_bits(acc[mt][nt][4 * g4 + 0]) * u : relu_bits(acc[mt][nt][4 * g4 + 0]), SC::SCALED ? relu_bits(acc[mt][nt][4 * g4 + 1]) * u : relu_bits(acc[mt][nt][4 * g4 + 1]),
                                   SC::SCALED ? relu_bits(acc[mt][nt][4 * g4 + 2]) * u : relu_bits(acc[mt][nt][4 * g4 + 2]), SC::SCALED ? relu_bits(acc[mt][nt][4 * g4 + 3]) * u : relu_bits(acc[mt][nt][4 * g4 + 3]));
                    }
            }
            if (SC::SCALED && ovf) { sovf[fpar] = 1; ovf = 0; }
        }
        // ---------------- layer 2 (stride 2): 4 output parities, oh = 2 ih - 1 + kh (the tap order of ConvT2Addr, mfma_pipe.h) ---------------
        float* Y = a.y2 + (size_t)img * (32 * 32 * 64);
        int T = 9;
#pragma unroll 1
        for (int par = 0; par < 4; ++par) {
            const int ph = par >> 1, pw = par & 1;
            acc_init(16, a.w2s);
            const int ntaps = (1 + ph) * (1 + pw);
#pragma unroll 1
            for (int t = 0; t < ntaps; ++t, ++T) {
                const int th = t / (1 + pw), tw = t - th * (1 + pw);
                const int da = (ph && th == 0) ? 1 : 0, db = (pw && tw == 0) ? 1 : 0;
                int pb[NTW];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int sy = prow0 + 2 * nt + da, sx = pcol + db;
                    pb[nt] = ((sy < 16 && sx < 16) ? sy * 16 + sx : 256) * DA3_PXB + g * 16;
                }
                const int npar = t + 1 < ntaps ? par : par + 1, nt_ = t + 1 < ntaps ? t + 1 : 0;
                tap(T, pb, T < 17 ? W2 + (size_t)l2_wt(npar, nt_) * (DA3_SLAB / 16) : nullptr);
            }
            if (live) {
                const bool ibad = SC::SCALED && sovf[fpar] != 0;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    float4* yp = reinterpret_cast<float4*>(Y) + (size_t)par * 4096 + ((2 * NTW * w + 2 * nt) * 16 + j) * 2 + g;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            float4 v;
                            v.x = relu_bits(acc[mt][nt][4 * g4 + 0]); v.y = relu_bits(acc[mt][nt][4 * g4 + 1]);
                            v.z = relu_bits(acc[mt][nt][4 * g4 + 2]); v.w = relu_bits(acc[mt][nt][4 * g4 + 3]);
                            if (SC::SCALED) { v.x *= a.w2s_inv; v.y *= a.w2s_inv; v.z *= a.w2s_inv; v.w *= a.w2s_inv; }
                            if (SC::SCALED && ibad) v.x = v.y = v.z = v.w = __builtin_inff();      // x4 or y1 did not fit fp16: the image is poisoned (k_dec_b_b3 turns it into a NaN sum)
                            yp[(mt * 4 + g4) * 512] = v;
                        }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_dec_b_b3: ConvTranspose2d(64, 32, s2) + ReLU + ConvTranspose2d(32, 1) + Sigmoid + the per-image sums (k_dec_b4 of decoder.hip;
// /root/reference/src/torchmodel.py:124-127, torchutils.py:26-37) with the 64 -> 32 contraction on the bf16 pipe, both operands as three
// bf16 planes.  Same input-stationary idea as k_dec_b4 -- the nine taps of the stride-2 layer read four shifted views of an input row --
// re-cut for operands that both come from LDS:
//   * one workgroup of EIGHT waves per CU and image, strips of SR = 4 input rows; wave (r, ph) = (w & 3, w >> 2) owns input row r of the
//     strip and the two output parities (ph, 0), (ph, 1): 3 taps for ph = 0, 6 for ph = 1.  Waves w and w + 4 share a SIMD (the hardware
//     deals a workgroup's waves to the SIMDs cyclically), so every SIMD carries 9 tap-tiles per 16-channel step like k_dec_b4's waves;
//   * the strip (5 rows incl. the halo) is split into its planes while it is staged: [pixel][3 planes][64 channels] bf16 + 16 B of padding
//     (400 B per pixel: conflict-free ds_read_b128 fragments, as in k_dec_a_b3);
//   * the weights are packed [16-channel step][tap][plane][64 lanes][8 bf16]: ONE 27 KiB slab per step holds every tap, copied global ->
//     LDS by LDS-DMA into two buffers (slab ks + 1 during step ks; 108 KiB per strip from L2 instead of 27 fragment loads per wave and step
//     -- 62 B/clk and CU, the whole L1 fill rate);
//   * per step a wave issues its taps as PAIRS on different accumulators (a third accumulator takes the odd tap so that no two
//     consecutive MFMAs share a chain), each next fragment read between two MFMAs;
//   * everything behind the contraction is k_dec_b4's: ReLU, the 32 -> 1 conv as v_mfma_f32_4x4x1 tap planes on the accumulators (the
//     C / D layout of the 32x32 bf16 and fp32 MFMAs is the same), horizontal pre-sums by DPP, H planes through an LDS ring, the gather of a
//     strip deferred into the next strip's contraction.
// LDS: strip 64.5 KB + two slabs 55.3 KB + H ring 27.6 KB = 147 KB.  Only launches of > 128 images take this kernel (the small, split
// launches stay on k_dec_b4<4>); per-image sums are reduced wave-wise, (((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7))).
// ---------------------------------------------------------------------------------------------------------
constexpr int DBB_ZPX = 5 * 32;                          // the zero pixel behind the five rows
constexpr int DBB_YROWS = 4 * 4 + 2;
template <class SC> struct DbbL {                        // LDS layout for NPL planes
    static constexpr int PXB = SC::NPL * 128 + 16;                       // bytes per staged pixel
    static constexpr int W0 = ((DBB_ZPX + 1) * PXB + 255) & ~255;
    static constexpr int SLAB = 9 * SC::NPL * 1024;                      // one 16-channel step: [9 taps][NPL planes][64 lanes][16 B]
    static constexpr bool RES = SC::NPL == 2;                            // two planes: all four steps' weights stay in LDS (72 KiB), copied once per workgroup
    static constexpr int H = W0 + (RES ? 4 : 2) * SLAB;
    static constexpr size_t LDS = H + (size_t)DBB_YROWS * 2 * 3 * 64 * sizeof(float);
};

// The four 16-channel steps of a strip for wave (r, PH).  abase: weight buffer 0 + lane * 16 (step ks in buffer ks & 1); bv[v]: the wave's
// four views (pixel byte address + g * 16).  sync(ks) = "slab ks has landed for everybody" (drain + barrier + the next slab's DMA request).
// Per step the taps go as PAIRS on different accumulators (a third accumulator takes the odd tap: no two consecutive MFMAs share a chain):
//   PH = 1: (7 -> 0 | 8 -> 1) on view 0, (1 -> 0 | 2 -> 1) on view 2, (6 -> 1 on view 1 | 0 -> 2 on view 3)
//   PH = 0: (4 -> 0 | 5 -> 1) on view 0, (3 on view 1, its products alternating between accumulators 1 and 2)
// (packed tap index kh * 3 + kw; views 0 = (r, j)  1 = (r, j + 1)  2 = (r + 1, j)  3 = (r + 1, j + 1)).  The fragments of a pair are read
// between the MFMAs of the pair before it -- ACROSS the step boundary too: sync(ks + 1) sits in front of the LAST pair of step ks, whose
// fragments are in registers by then, so the first pair of a step never starts with a burst of twelve exposed reads (that cost 12 % of the
// kernel: profiles/r6_dec_b_b3_ablation.txt).  Products in the order (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi): small terms first.
template <class SC, int PH, class Sync>
__device__ __forceinline__ void dbb_strip(f32x16 (&acc)[3], const unsigned char* abase, const unsigned char* const (&bv)[4], Sync sync) {
    constexpr int NPL = SC::NPL, NMF = 2 * SC::NPR, NRD = 4 * NPL, SLAB = DbbL<SC>::SLAB;
    constexpr int NP = PH ? 3 : 2;
    constexpr int TX[3] = {PH ? 7 : 4, PH ? 1 : 3, 6}, TY[3] = {PH ? 8 : 5, PH ? 2 : 3, 0};
    constexpr int VX[3] = {0, PH ? 2 : 1, 1}, VY[3] = {0, PH ? 2 : 1, 3};
    constexpr int AX[3] = {0, PH ? 0 : 1, 1}, AY[3] = {1, PH ? 1 : 2, 2};
    float4 ax[2][NPL], ay[2][NPL], bx[2][NPL], by[2][NPL];
    auto ld = [&](int ks, int i, int l, float4 (&axn)[NPL], float4 (&ayn)[NPL], float4 (&bxn)[NPL], float4 (&byn)[NPL]) {       // fragment read l of pair i of step ks
        const unsigned char* slab = abase + (DbbL<SC>::RES ? ks : (ks & 1)) * SLAB;
        const int k = l / NPL, p = l % NPL;
        if (k == 0) axn[p] = *reinterpret_cast<const float4*>(slab + (TX[i] * NPL + p) * 1024);
        else if (k == 1) bxn[p] = *reinterpret_cast<const float4*>(bv[VX[i]] + p * 128 + ks * 32);
        else if (k == 2) { if (TY[i] != TX[i]) ayn[p] = *reinterpret_cast<const float4*>(slab + (TY[i] * NPL + p) * 1024); }
        else { if (VY[i] != VX[i]) byn[p] = *reinterpret_cast<const float4*>(bv[VY[i]] + p * 128 + ks * 32); }
    };
    sync(0);
#pragma unroll
    for (int l = 0; l < NRD; ++l) ld(0, 0, l, ax[0], ay[0], bx[0], by[0]);
#pragma unroll
    for (int q = 0; q < 4 * NP; ++q) {
        const int i = q % NP, c = q & 1, n = c ^ 1;
        const int qn = q + 1, ksn = qn / NP, in_ = qn % NP;
        const bool more = qn < 4 * NP;
        if (more && in_ == 0) sync(ksn);
        const bool sameT = TY[i] == TX[i], sameV = VY[i] == VX[i];
#pragma unroll
        for (int m = 0; m < NMF; ++m) {
            const int pr = m >> 1;
            if (!sameT) {
                if ((m & 1) == 0) acc[AX[i]] = SC::mfma(ax[c][SC::PA(pr)], bx[c][SC::PB(pr)], acc[AX[i]]);
                else acc[AY[i]] = SC::mfma(ay[c][SC::PA(pr)], (sameV ? bx[c][SC::PB(pr)] : by[c][SC::PB(pr)]), acc[AY[i]]);
            } else if ((m & 1) == 0) {                        // one tap, its products alternating between two accumulators
                if (pr & 1) acc[AY[i]] = SC::mfma(ax[c][SC::PA(pr)], bx[c][SC::PB(pr)], acc[AY[i]]);
                else acc[AX[i]] = SC::mfma(ax[c][SC::PA(pr)], bx[c][SC::PB(pr)], acc[AX[i]]);
            }
            if (more) {
#pragma unroll
                for (int l = m * NRD / NMF; l < (m + 1) * NRD / NMF; ++l) ld(ksn, in_, l, ax[n], ay[n], bx[n], by[n]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <class SC>
__global__ void __launch_bounds__(512, 1) k_dec_b_b3(const DecBArgs a) {
    constexpr int SR = 4, NW = 8, NTHR = 512, NS = 32 / SR, NPF = (SR + 1) * 512 / NTHR, NPL = SC::NPL;
    constexpr int DBB_PXB = DbbL<SC>::PXB, DBB_W0 = DbbL<SC>::W0, DBB_SLAB = DbbL<SC>::SLAB, DBB_H = DbbL<SC>::H;
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    float* sH = reinterpret_cast<float*>(smb + DBB_H);          // [ring row][channel half][kh][64 output columns]
    __shared__ float4 sb3[8];
    __shared__ float4 sW4[8 * 12];
    __shared__ float sred[2][NW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int r = w & 3, ph = w >> 2;
    // persistent over images (one workgroup per CU: nothing else hides a workgroup's start-up): static stride, dead rows of the call skipped
    auto next_live = [&](int i) { while (i < a.rows && !row_live(a.live, i)) i += (int)gridDim.x; return i; };
    int img = next_live((int)blockIdx.x);
    if (img >= a.rows) return;

    if (tid < 8) sb3[tid] = reinterpret_cast<const float4*>(a.b3)[tid];
    if (tid < 96) {                                              // the 4x4x1 tap table of k_dec_b4: pattern = h * 4 + kw, float4 i4 = (kh, g4)
        const int pat = tid / 12, i4 = tid - pat * 12;
        const int kw = pat & 3, hh = pat >> 2, kh = i4 >> 2, g4 = i4 & 3;
        float4 q4 = kw < 3 ? reinterpret_cast<const float4*>(a.w4 + (3 * kh + kw) * 32)[2 * g4 + hh] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (SC::SCALED) { q4.x *= a.w3s_inv; q4.y *= a.w3s_inv; q4.z *= a.w3s_inv; q4.w *= a.w3s_inv; }      // ReLU commutes with the (positive, power-of-two) weight scale: undone in the taps
        sW4[tid] = q4;
    }
    const float4* w4p = sW4 + (h * 4 + (lane & 3)) * 12;
    if (tid < DBB_PXB / 4) reinterpret_cast<uint32_t*>(smb + DBB_ZPX * DBB_PXB)[tid] = 0u;

    const f32x4* Y2 = reinterpret_cast<const f32x4*>(a.y2);
    auto y2_at = [&](int iy, int idx) -> size_t { return (size_t)(((iy & 1) * 16 + ((idx & 511) >> 5)) * 512 + (iy >> 1) * 32 + (idx & 31)); };
    f32x4 pf[NPF];
#pragma unroll
    for (int it = 0; it < NPF; ++it) pf[it] = Y2[(size_t)img * (32 * 32 * 16) + y2_at(it, tid)];
    // weight slabs by LDS-DMA: piece i (1 KiB) of a slab is copied by wave i & 7
    const char* Wg = reinterpret_cast<const char*>(a.w3b3) + lane * 16;
    const unsigned lds_w0 = (unsigned)(size_t)(smb + DBB_W0);
    auto slab_dma = [&](int ks, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = w + 8 * i;
            if (piece < 9 * NPL) glds16(Wg + (size_t)ks * DBB_SLAB + piece * 1024, lds_w0 + (unsigned)(buf * DBB_SLAB + piece * 1024));
        }
    };
    constexpr bool RES = DbbL<SC>::RES;
    if (RES) { slab_dma(0, 0); slab_dma(1, 1); slab_dma(2, 2); slab_dma(3, 3); }      // resident weights: no per-step copies, no per-step barriers
    else slab_dma(0, 0);
    uint32_t ovf = 0;                                    // fp16 split: something this thread staged for the current image did not fit fp16
    auto put_planes = [&](unsigned char* px, int c, const float v0, const float v1, const float v2, const float v3) {
        uint32_t pa_[NPL], pb_[NPL];
        SC::split_pk(v0, v1, pa_); SC::split_pk(v2, v3, pb_);
        ovf |= SC::overflow(pa_[0]) | SC::overflow(pb_[0]);
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<uint2*>(px + p * 128 + c * 2) = make_uint2(pa_[p], pb_[p]);
    };
    // the four shifted views of this wave's input row (column 32 does not exist: the zero pixel)
    const int spx[4] = {r * 32 + j, (j < 31) ? r * 32 + j + 1 : DBB_ZPX, (r + 1) * 32 + j, (j < 31) ? (r + 1) * 32 + j + 1 : DBB_ZPX};
    const unsigned char* abase = smb + DBB_W0 + lane * 16;

    const float D1 = 1.00001f, D0 = 0.00001f;
    int hb = 0, hbp = 0, nimgs = 0;
#pragma unroll 1
    for (; img < a.rows; ++nimgs) {
        const int nimg = next_live(img + (int)gridDim.x);
        const int mg = a.m0 + img;
        const int g = mg / a.rows_per_group;
        const int rr = mg - g * a.rows_per_group;
        int gt, gp, gs;
        group_decode(a.gm, g, gt, gp, gs);
        const int mode = (gp == 0 && a.reward0) ? 1 : 0;
        const int slot = (gp == 0 && a.store0) ? gt * a.gm.S + gs : -1;
        float* po = (slot >= 0) ? a.po + ((size_t)slot * a.rows_per_group + rr) * 4096 : nullptr;
        float part = 0.f;
        float gh[2][6], gpr[2] = {0.f, 0.f};
        // ---- gather of output row oh (lane = column), k_dec_b4's: out = b4 + H[0][oh + 1] + H[1][oh] + H[2][oh - 1], channel halves added here
        auto g_load = [&](int oh, int q, int r0y, int hb0) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int tr = oh + 1 - kh;
                const bool rv = tr >= 0 && tr <= 63;
                int hs = hb0 + (tr - r0y);
                hs = hs < 0 ? hs + DBB_YROWS : hs;
                hs = hs >= DBB_YROWS ? hs - DBB_YROWS : hs;
                hs = rv ? hs : 0;
                const float* hq = sH + ((hs * 2) * 3 + kh) * 64 + lane;
                gh[q][2 * kh] = hq[0]; gh[q][2 * kh + 1] = hq[3 * 64];
            }
        };
        auto g_sig = [&](int oh, int q) {
            float v = a.b4;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int tr = oh + 1 - kh;
                v += (tr >= 0 && tr <= 63) ? gh[q][2 * kh] + gh[q][2 * kh + 1] : 0.f;
            }
            gpr[q] = hw_sigmoid(v);
        };
        auto g_term = [&](int oh, int q) {
#pragma clang fp contract(off)
            const float pr = gpr[q];
            const float l1 = hw_log(D1 - pr), l0 = hw_log(D0 + pr);
            const float te = __builtin_fmaf(pr - 1.0f, l1, -(pr * l0));
            const float tw = reward_term(pr, oh, lane, 64, 64, a.reward_intent);
            const float t = mode == 0 ? te : tw;
            part += oh >= 0 ? t : 0.0f;
        };
        auto g_store = [&](int oh, int q) {
            if (po && oh >= 0) {
                int owl = lane; asm volatile("" : "+v"(owl));
                (po + oh * 64)[owl] = gpr[q];
            }
        };
#pragma unroll 1
        for (int s = 0; s < NS; ++s) {
            const int oh0 = 2 * SR * (s - 1) - 1 + r, oh1 = oh0 + 4;      // the light waves' two output rows of the previous strip (deferred gather)
            // ---- stage the strip: thread -> (row it, pixel ix, channel quad c4) of k_dec_b4's y2 addressing, split into planes
#pragma unroll
            for (int it = 0; it < NPF; ++it) {
                const int seg = tid >> 5, wi = tid & 31;
                const int ix = 2 * (wi >> 1) + (seg >> 3), c4 = 2 * (seg & 7) + (wi & 1);
                const bool in = SR * s + it < 32;
                put_planes(smb + (size_t)(it * 32 + ix) * DBB_PXB, 4 * c4, in ? pf[it][0] : 0.f, in ? pf[it][1] : 0.f, in ? pf[it][2] : 0.f, in ? pf[it][3] : 0.f);
            }
            f32x16 acc[3];
            const bool last = s + 1 == NS;
            auto sync = [&](int ks) {
                if (!RES || ks == 0) {
                    glds_drain();                                 // this wave's pieces of slab ks have landed ...
                    __syncthreads();                              // ... and everybody's (ks = 0: the staged strip too); nobody reads the other buffer any more
                }
                if (ks == 0) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {              // accumulators start at the bias (register e holds channel (e & 3) + 8 (e >> 2) + 4 h)
                        float4 bb = sb3[2 * g4 + h];
                        if (SC::SCALED) { bb.x *= a.w3s; bb.y *= a.w3s; bb.z *= a.w3s; bb.w *= a.w3s; }       // the accumulators hold (weight scale) x y3
                        acc[0][4 * g4] = bb.x; acc[0][4 * g4 + 1] = bb.y; acc[0][4 * g4 + 2] = bb.z; acc[0][4 * g4 + 3] = bb.w;
                    }
                    acc[1] = acc[0];
                    acc[2] = (f32x16)(0.f);
                }
                if (!RES && (ks < 3 || !last || nimg < a.rows)) slab_dma((ks + 1) & 3, (ks + 1) & 1);
                if (ks == 3) {                                    // the next strip's rows (the next image's first strip behind the last one), behind the slab request
                    const size_t ib = (size_t)(last ? (nimg < a.rows ? nimg : img) : img) * (32 * 32 * 16);
                    const int r0 = last ? 0 : SR * (s + 1);
#pragma unroll
                    for (int it = 0; it < NPF; ++it) pf[it] = Y2[ib + y2_at(min(r0 + it, 31), tid)];
                }
                // the previous strip's gather, a piece per step, on the LIGHT waves (ph = 0: half the matrix work of their SIMD partners)
                if (s > 0 && ph == 0) {
                    if (ks == 1) { g_load(oh0, 0, 2 * SR * (s - 1), hbp); g_load(oh1, 1, 2 * SR * (s - 1), hbp); }
                    if (ks == 2) { g_sig(oh0, 0); g_term(oh0, 0); }
                    if (ks == 3) { g_sig(oh1, 1); g_term(oh1, 1); }
                }
            };
            const unsigned char* bv[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) bv[v] = smb + (size_t)spx[v] * DBB_PXB + h * 16;
            if (ph) dbb_strip<SC, 1>(acc, abase, bv, sync); else dbb_strip<SC, 0>(acc, abase, bv, sync);
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[1][e] += acc[2][e];
            // ---- ReLU, then the 32 -> 1 conv as tap planes on the accumulators (k_dec_b4's tail for this wave's row parity ph)
            float w4g[3][16];
#pragma unroll
            for (int i4 = 0; i4 < 12; ++i4) {
                const float4 q = w4p[i4];
                w4g[i4 >> 2][4 * (i4 & 3)] = q.x; w4g[i4 >> 2][4 * (i4 & 3) + 1] = q.y; w4g[i4 >> 2][4 * (i4 & 3) + 2] = q.z; w4g[i4 >> 2][4 * (i4 & 3) + 3] = q.w;
            }
            {
                f32x4 Tq[2][3];
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc[0][e] = relu_bits(acc[0][e]); acc[1][e] = relu_bits(acc[1][e]); }
#pragma unroll
                for (int pw = 0; pw < 2; ++pw)
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) Tq[pw][kh] = (f32x4)(0.f);
#pragma unroll
                for (int e = 0; e < 16; ++e)
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        Tq[0][kh] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4g[kh][e], acc[0][e], Tq[0][kh], 0, 0, 0);
                        Tq[1][kh] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4g[kh][e], acc[1][e], Tq[1][kh], 0, 0, 0);
                    }
                int hsw = hb + 2 * r + ph;                        // ring slot of this wave's y3 row 2 (SR s + r) + ph
                hsw = hsw >= DBB_YROWS ? hsw - DBB_YROWS : hsw;
                float* hp = sH + ((hsw * 2 + h) * 3) * 64 + 2 * j;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    float l = wave_shr1(Tq[1][kh][2]), rgt = wave_shl1(Tq[0][kh][0]);
                    l = (j == 0) ? 0.f : l;
                    rgt = (j == 31) ? 0.f : rgt;
                    float2 eo;
                    eo.x = (Tq[1][kh][0] + Tq[0][kh][1]) + l;
                    eo.y = (rgt + Tq[1][kh][1]) + Tq[0][kh][2];
                    *reinterpret_cast<float2*>(hp + kh * 64) = eo;
                }
            }
            __syncthreads();
            if (s > 0 && ph == 0) { g_store(oh0, 0); g_store(oh1, 1); }
            if (s == NS - 1) {                                    // the last strip's own rows (and row 63 behind it)
#pragma unroll
                for (int q = 0; q <= 1; ++q) {
                    if (q == 1 && w != 0) break;
                    const int oh = 2 * SR * s - 1 + q * NW + w;
                    g_load(oh, 0, 2 * SR * s, hb); g_sig(oh, 0); g_term(oh, 0); g_store(oh, 0);
                }
            }
            hbp = hb;
            hb += 2 * SR;
            hb = hb >= DBB_YROWS ? hb - DBB_YROWS : hb;
        }
        // the image's sum: wave-wise, then (((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7))); the partials alternate between two LDS rows, so
        // the only barrier is the one behind the NEXT image's first staging (the last image: an own barrier)
        float v = part;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (SC::SCALED) {                                 // an input of this image did not fit fp16 (or arrived poisoned): the sum is NaN, never a finite wrong number
            if (__any((int)(ovf != 0))) v = __builtin_nanf("");
            ovf = 0;
        }
        if (lane == 0) sred[nimgs & 1][w] = v;
        __syncthreads();
        {
            const float* q = sred[nimgs & 1];
            const float tot = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
            if (tid == 0) a.val[mg] = tot;
            if (SC::SCALED && po && tot != tot) {           // a poisoned image that is also STORED (the D1 pass, efe_decoder): the stored pixels say so too
#pragma unroll
                for (int i = 0; i < 4096 / NTHR; ++i) po[i * NTHR + tid] = tot;
            }
        }
        img = nimg;
    }
}

void launch_dec_b_b3(const DecBArgs& a, hipStream_t st) {
    const int grid = a.rows < 256 ? a.rows : 256;        // persistent: one workgroup per CU
    if (a.split == 2) hipLaunchKernelGGL(k_dec_b_b3<SchH2>, dim3(grid), dim3(512), DbbL<SchH2>::LDS, st, a);
    else hipLaunchKernelGGL(k_dec_b_b3<SchB3>, dim3(grid), dim3(512), DbbL<SchB3>::LDS, st, a);
}

// ConvT3 weights for k_dec_b_b3: [Cin = 64][Cout = 32][3][3] -> [ks = Cin / 16][tap][NPL planes][64 lanes][8 x 16 bit]; returns the weight scale
float pack_convt3_split(int mode, const float* W_cicokk, uint16_t* dst) {
    const int npl = mode == 2 ? 2 : 3;
    const float sc = split_weight_scale(mode, W_cicokk, (size_t)64 * 32 * 9);
    for (int ks = 0; ks < 4; ++ks)
        for (int t = 0; t < 9; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 8; ++s) {
                    const int co = lane & 31, ci = ks * 16 + 8 * (lane >> 5) + s;
                    uint32_t p[3];
                    split_host(mode, W_cicokk[((size_t)ci * 32 + co) * 9 + t] * sc, p);
                    for (int pl = 0; pl < npl; ++pl) dst[((((size_t)ks * 9 + t) * npl + pl) * 64 + lane) * 8 + s] = (uint16_t)p[pl];
                }
    return sc;
}

void launch_dec_a_b3(const DecAArgs& a, hipStream_t st) {
    const int grid = a.rows < 256 ? a.rows : 256;        // persistent: one workgroup per CU
    if (a.split == 2) hipLaunchKernelGGL((k_dec_a_b3<SchH2, EFE_DA3_NTW>), dim3(grid), dim3(512 / EFE_DA3_NTW), Da3L<2>::LDS, st, a);
    else hipLaunchKernelGGL((k_dec_a_b3<SchB3, EFE_DA3_NTW>), dim3(grid), dim3(512 / EFE_DA3_NTW), Da3L<3>::LDS, st, a);
}

// conv weights for k_dec_a_b3: get(tap, co, ci) -> [tap][2 mt][4 ks][NPL planes][64 lanes][8 x 16 bit]; returns the weight scale
float pack_conv_split(int mode, const float* W_cicokk, int Cin, int Cout, uint16_t* dst) {
    const int npl = mode == 2 ? 2 : 3;
    const float sc = split_weight_scale(mode, W_cicokk, (size_t)Cin * Cout * 9);
    for (int t = 0; t < 9; ++t)
        for (int mt = 0; mt < Cout / 32; ++mt)
            for (int ks = 0; ks < Cin / 16; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 8; ++s) {
                        const int co = mt * 32 + (lane & 31), ci = ks * 16 + 8 * (lane >> 5) + s;
                        uint32_t p[3];
                        split_host(mode, W_cicokk[((size_t)ci * Cout + co) * 9 + t] * sc, p);       // ConvTranspose2d weights are [Cin][Cout][kh][kw]
                        for (int pl = 0; pl < npl; ++pl)
                            dst[(((((size_t)t * (Cout / 32) + mt) * (Cin / 16) + ks) * npl + pl) * 64 + lane) * 8 + s] = (uint16_t)p[pl];
                    }
    return sc;
}

int init_bf16x3_kernels() {
    if (hipFuncSetAttribute((const void*)k_dec_a_b3<SchB3, EFE_DA3_NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Da3L<3>::LDS) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_dec_a_b3<SchH2, EFE_DA3_NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Da3L<2>::LDS) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_dec_b_b3<SchB3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DbbL<SchB3>::LDS) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_dec_b_b3<SchH2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DbbL<SchH2>::LDS) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_fc4_b3<SchH2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Fc4L<2>::LDS) != hipSuccess) return 1;
    return hipFuncSetAttribute((const void*)k_fc4_b3<SchB3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Fc4L<3>::LDS) != hipSuccess;
}

void launch_fc4_b3(const GemmArgs& a, hipStream_t st) {
    // persistent: one 8-wave workgroup per CU, 8 feature groups x 32 workgroups, each with >= 1 (row tile, step) pair
    const int rt = a.split == 2 ? 32 * Fc4L<2>::NT : 32 * Fc4L<3>::NT;
    const int nsteps = ((a.n_pix + rt - 1) / rt) * fc4b3_spg(a.mtiles);
    const int nper = nsteps < 32 ? nsteps : 32;
    if (a.split == 2) hipLaunchKernelGGL(k_fc4_b3<SchH2>, dim3(8 * nper), dim3(512), Fc4L<2>::LDS, st, a);
    else hipLaunchKernelGGL(k_fc4_b3<SchB3>, dim3(8 * nper), dim3(512), Fc4L<3>::LDS, st, a);
}

// host side of efe_commit_weights: W [out][in = 256] (rows already in the engine's NHWC feature order through row_perm) -> NPL 16-bit
// planes, fragment-major [out / 32][16][NPL][64 lanes][8]; returns the weight scale (1 for the bf16 split)
float pack_dense_split(int mode, const float* W, const int* row_perm, int out, int in, uint16_t* dst) {
    const int mtiles = out / 32, npl = mode == 2 ? 2 : 3;
    const float sc = split_weight_scale(mode, W, (size_t)out * in);
    for (int mt = 0; mt < mtiles; ++mt)
        for (int ks = 0; ks < in / 16; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 8; ++s) {
                    const int co = mt * 32 + (lane & 31), ci = ks * 16 + 8 * (lane >> 5) + s;
                    uint32_t p[3];
                    split_host(mode, W[(size_t)(row_perm ? row_perm[co] : co) * in + ci] * sc, p);
                    for (int pl = 0; pl < npl; ++pl) dst[((((size_t)mt * (in / 16) + ks) * npl + pl) * 64 + lane) * 8 + s] = (uint16_t)p[pl];
                }
    return sc;
}

}  // namespace efe
