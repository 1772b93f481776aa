// Shared MFMA pipeline pieces of the LDS-staged kernels (decoder.hip, encoder.hip): fp32 32x32x2 MFMA macros and the
// software-pipelined tap contraction (TapPipe) over LDS images (padded pixel slots; k_fc4's batch tile is XOR-swizzled).
#pragma once
#include "kernels.h"

#ifndef EFE_PD_WIDE
#define EFE_PD_WIDE 1      // 2x2 tiles: 16 MFMAs = 1024 cycles per chunk
#endif
#ifndef EFE_PD_NARROW
#define EFE_PD_NARROW 2    // 1x2 / 1x1 tiles: 8 / 4 MFMAs per chunk
#endif

namespace efe {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));     // native vector: plain loads/stores, no struct memcpy

#define MFMA4(ACC, AV, BV)                                                      \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).x, (BV).x, ACC, 0, 0, 0);   \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).y, (BV).y, ACC, 0, 0, 0);   \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).z, (BV).z, ACC, 0, 0, 0);   \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).w, (BV).w, ACC, 0, 0, 0);

// the same with the chain's start value INIT (e.g. the bias vector) as the C operand of the first instruction
#define MFMA4I(ACC, INIT, AV, BV)                                               \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).x, (BV).x, INIT, 0, 0, 0);  \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).y, (BV).y, ACC, 0, 0, 0);   \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).z, (BV).z, ACC, 0, 0, 0);   \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).w, (BV).w, ACC, 0, 0, 0);

// Weight (A) fragments are fetched with buffer loads: SGPR resource (base) + SGPR byte offset (tap / tile / chunk, all
// uniform) + ONE per-lane VGPR offset (lane * 16).  With flat global loads hipcc folds the lane into a 64-bit per-lane
// pointer and materialises a VGPR pair per distinct offset -- dozens of pairs, hoisted out of the loops and spilled.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wrsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);   // raw buffer, 2 GiB window
}
__device__ __forceinline__ float4 wfrag(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, size_t f4_index) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, (unsigned)f4_index * 16u, 0);
    return __builtin_bit_cast(float4, v);
}

// max(x, 0) for fp32 as a signed-integer max of the bit pattern: negative floats (and -0, -NaN) are negative integers.
__device__ __forceinline__ float relu_bits(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// sigmoid and natural log on the hardware transcendental instructions alone (v_exp_f32 = 2^x, v_rcp_f32, v_log_f32 = log2 x: 1 ulp each).
// __expf / __logf add a denormal-range rescue (compare + ldexp + select per call) that these call sites do not need: the log arguments
// are d + p and (d + 1) - p with p in [0, 1], d = 1e-5; 2^x underflowing to 0 or overflowing to inf gives the sigmoid's limits 1 and 0.
__device__ __forceinline__ float hw_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }
__device__ __forceinline__ float hw_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

// DPP row shifts of one fp32 register inside 16-lane rows (measured semantics, tools/ubench/dpp_probe.hip): shr1: lane i <- lane i - 1,
// shl1: lane i <- lane i + 1; lanes without a source are zero (_zero) or keep `old` (_keep); ror1: lane i <- lane (i - 1) & 15, ror15: <- (i + 1) & 15
__device__ __forceinline__ float dpp_shr1_zero(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, true)); }
__device__ __forceinline__ float dpp_shl1_zero(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x101, 0xf, 0xf, true)); }
__device__ __forceinline__ float dpp_shr1_keep(float old, float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, false)); }
__device__ __forceinline__ float dpp_shl1_keep(float old, float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), 0x101, 0xf, 0xf, false)); }
__device__ __forceinline__ float wave_shr1(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, true)); }   // lane i <- i - 1 over the wave
__device__ __forceinline__ float wave_shl1(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, true)); }   // lane i <- i + 1
__device__ __forceinline__ float dpp_ror1(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false)); }
__device__ __forceinline__ float dpp_ror15(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x12f, 0xf, 0xf, false)); }



// Software-pipelined contraction over `ntaps` taps x 64 input channels (8 chunks of 8): the A fragments (weights,
// L2) and B fragments (activations, LDS) of chunk i+1 are requested before the MFMAs of chunk i issue, so neither
// latency is exposed (hipcc otherwise waits for each chunk's loads right before its first MFMA).
// addr(t, base[NT], sw[NT], wtap): LDS float4 base + swizzle key of every pixel tile and the packed-weight tap index.
// Wl is the UNIFORM base of the packed weights: the fragment address is (scalar base + scalar tap/chunk offset) + lane, i.e.
// global_load with an SGPR base and one 32-bit lane offset.  A per-lane 64-bit pointer instead makes hipcc materialise a
// VGPR pair per distinct offset (dozens, hoisted out of the loops and spilled).
struct ConvWIdx {          // packed conv weights [tap][MT tiles][8 chunks][64 lanes]
    template <int MT> __device__ __forceinline__ static size_t at(int wt, int mt, int kc) { return (size_t)((wt * MT + mt) * 8 + kc) * 64; }
};
struct DenseWIdx {         // packed dense weights [feature tile][32 chunks][64 lanes]; "tap" t = 64-channel slice of K = 256
    int mt0;
    template <int MT> __device__ __forceinline__ size_t at(int wt, int mt, int kc) const { return (size_t)((mt0 + mt) * 32 + wt * 8 + kc) * 64; }
};

// tap -> LDS source of the stride-2 transposed convs (oh = 2*ih - 1 + kh: even output rows use kh=1 (ih=a); odd rows use
// kh=0 (ih=a+1) and kh=2 (ih=a)); `rows`/`cols` bound the staged image, `zero` is the zero-pixel slot.
// PS = float4 slots per staged pixel: 16 = XOR-swizzled quads (swz), 17 = padded slots (no swizzle key)
template <int NT, int PS = 16>
struct ConvT2Addr {
    int ph, pw, row0, row_step, col, rows, cols, zero;
    __device__ __forceinline__ void operator()(int t, int (&bs)[NT], int (&sw)[NT], int& wt) const {
        const int th = t / (1 + pw), tw = t - th * (1 + pw);
        const int kh = ph ? (th ? 2 : 0) : 1, da = (ph && th == 0) ? 1 : 0;
        const int kw = pw ? (tw ? 2 : 0) : 1, db = (pw && tw == 0) ? 1 : 0;
        wt = kh * 3 + kw;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int sy = row0 + row_step * nt + da, sx = col + db;
            const int sp = (sy < rows && sx < cols) ? sy * cols + sx : zero;
            bs[nt] = sp * PS; sw[nt] = PS == 16 ? (sp & 15) : 0;
        }
    }
};

// PD = prefetch distance of the weight (A) fragments in chunks: they come from L2 (~500-900 cycles under load), and a
// chunk of MT*NT*4 MFMAs only covers MT*NT*256 cycles, so narrow tiles need PD >= 2.  B fragments (LDS) stay 1 ahead.
template <int MT, int NT, int KC = 8, int PD = 1>
struct TapPipe {
    static_assert(KC % PD == 0, "the A-fragment ring must wrap consistently across taps");
    float4 aq[PD][MT];           // A fragments of the next PD chunks (already requested)
    float4 bv[NT];               // B fragments of the next chunk
    int bs[NT], sw[NT], wt;

    // request the first chunks of a contraction; call it as early as the operands are valid
    template <class AddrFn, class WIdx>
    __device__ __forceinline__ void begin(const float4* __restrict__ Wl, const float4* sm, const int h, AddrFn addr, WIdx widx) {
        const unsigned ln = (threadIdx.x & 63u) * 16u;
        const __amdgpu_buffer_rsrc_t wr = wrsrc(Wl);
        addr(0, bs, sw, wt);
#pragma unroll
        for (int p = 0; p < PD; ++p)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) aq[p][mt] = wfrag(wr, ln, widx.template at<MT>(wt, mt, p));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = sm[bs[nt] + (h ^ sw[nt])];
    }

    template <class AddrFn, class WIdx>
    __device__ __forceinline__ void run(f32x16 (&acc)[MT][NT], const int ntaps, const float4* __restrict__ Wl,
                                        const float4* sm, const int h, AddrFn addr, WIdx widx) {
        const unsigned ln = (threadIdx.x & 63u) * 16u;
        const __amdgpu_buffer_rsrc_t wr = wrsrc(Wl);
        for (int t = 0; t < ntaps; ++t) {
            int nbs[NT], nsw[NT], nwt;
            addr((t + 1 < ntaps) ? t + 1 : t, nbs, nsw, nwt);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                float4 av[MT], bn[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt] = aq[kc % PD][mt];
                // chunk kc + PD of this tap, or chunk kc + PD - KC of the next one, replaces the slot just consumed
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    aq[kc % PD][mt] = (kc + PD < KC) ? wfrag(wr, ln, widx.template at<MT>(wt, mt, kc + PD))
                                                     : wfrag(wr, ln, widx.template at<MT>(nwt, mt, kc + PD - KC));
                if (kc < KC - 1) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bn[nt] = sm[bs[nt] + ((2 * (kc + 1) + h) ^ sw[nt])];
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bn[nt] = sm[nbs[nt] + (h ^ nsw[nt])];
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the prefetch loads AHEAD of this chunk's MFMAs (hipcc sinks them otherwise)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { MFMA4(acc[mt][nt], av[mt], bv[nt]) }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = bn[nt];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { bs[nt] = nbs[nt]; sw[nt] = nsw[nt]; }
            wt = nwt;
        }
    }
};

template <int MT, int NT, class AddrFn, class WIdx>
__device__ __forceinline__ void tap_loop(f32x16 (&acc)[MT][NT], const int ntaps, const float4* __restrict__ Wl,
                                         const float4* sm, const int h, AddrFn addr, WIdx widx) {
    TapPipe<MT, NT, 8, (MT * NT >= 4 ? EFE_PD_WIDE : EFE_PD_NARROW)> p;
    p.begin(Wl, sm, h, addr, widx);
    p.run(acc, ntaps, Wl, sm, h, addr, widx);
}


// explicit prefetch distance (kernels with 4 waves per SIMD and a tight VGPR budget use PD = 1 for narrow tiles too)
template <int MT, int NT, int PD, class AddrFn, class WIdx>
__device__ __forceinline__ void tap_loop_pd(f32x16 (&acc)[MT][NT], const int ntaps, const float4* __restrict__ Wl,
                                            const float4* sm, const int h, AddrFn addr, WIdx widx) {
    TapPipe<MT, NT, 8, PD> p;
    p.begin(Wl, sm, h, addr, widx);
    p.run(acc, ntaps, Wl, sm, h, addr, widx);
}

template <int MT, int NT, int KC, class AddrFn, class WIdx>
__device__ __forceinline__ void tap_loop_kc(f32x16 (&acc)[MT][NT], const int ntaps, const float4* __restrict__ Wl,
                                            const float4* sm, const int h, AddrFn addr, WIdx widx) {
    TapPipe<MT, NT, KC, EFE_PD_NARROW> p;
    p.begin(Wl, sm, h, addr, widx);
    p.run(acc, ntaps, Wl, sm, h, addr, widx);
}

// chunk count and prefetch distance both explicit
template <int MT, int NT, int KC, int PD, class AddrFn, class WIdx>
__device__ __forceinline__ void tap_loop_kc_pd(f32x16 (&acc)[MT][NT], const int ntaps, const float4* __restrict__ Wl,
                                               const float4* sm, const int h, AddrFn addr, WIdx widx) {
    TapPipe<MT, NT, KC, PD> p;
    p.begin(Wl, sm, h, addr, widx);
    p.run(acc, ntaps, Wl, sm, h, addr, widx);
}

// packed weights [tap][mtiles][kcn chunks][64 lanes] with a tile offset
struct PackedWIdx {
    int mtiles, kcn, mt0;
    template <int MT> __device__ __forceinline__ size_t at(int wt, int mt, int kc) const {
        return (size_t)((wt * mtiles + mt0 + mt) * kcn + kc) * 64;
    }
};

}  // namespace efe
