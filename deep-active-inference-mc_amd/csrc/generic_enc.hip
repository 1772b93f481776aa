// Encoder convolutions of the geometry-generic path, LDS-tiled: Conv2d(k3, s2, p0) + ReLU for the first two layers of
// /root/reference/src/torchmodel.py:85-88 at any resolution (BASELINE configs[4]: 3 x 84 x 84 -> 41 x 41 x 32 -> 20 x 20 x 32),
// which are 80 % of the generic encoder's time.  k_conv_g (generic.hip) read every operand straight from L2 -- each input element
// up to 9 / 4 times, no prefetch: MFMA-busy 0.33 at 5.3 TB/s (profiles/r3_v2_ai) -- and still serves layers 3 and 4.
//
// One workgroup per image walks down strips of TY output rows (<= 128 output pixels = one 32-pixel tile per wave).  The input rows
// 2 y0 .. 2 y0 + 2 TY of a strip live in LDS as a ring of 2 TY + 1 rows (consecutive strips share one row in place; the 2 TY new
// rows -- one contiguous block of global memory -- are prefetched into registers while the strip is contracted), with the columns of
// a row DE-INTERLEAVED (even columns first, then odd): tap kw of output column x reads input column 2 x + kw, i.e. slot x of the
// even half (kw 0), slot x of the odd half (kw 1) or slot x + 1 of the even half (kw 2), so consecutive lanes read consecutive
// slots.  Channels are the MFMA rows (A = packed weights), pixels its columns (B = the LDS strip): a lane ends with 16 channels of
// one pixel and stores four float4.
//   L = 2 (32 input channels): 9 float4 per slot (8 + 1 padding), 9 taps x 4 channel blocks through the shared TapPipe.
//   L = 1 (C <= 3 input channels read from the NHWC4 image): one float4 per slot holding (c0, c2, c1, c3), so that lane half h reads
//       the float2 (c_h, c_{2+h}) = its K operand of the tap's two MFMAs; weights as 18 resident registers.
#include "mfma_pipe.h"

namespace efe {

template <int L, int NPF>
__global__ void __launch_bounds__(256, 2) k_conv_e(const ConvEArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 sx[];        // [NR ring rows][Win slots][PS4]
    constexpr int CQ = L == 1 ? 1 : 8;           // float4 per input pixel that are staged
    constexpr int PS4 = L == 1 ? 1 : 9;          // float4 per LDS slot
    constexpr int IS = L == 1 ? GEN_IMG_LD : 32; // floats per input pixel in global memory
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int img = blockIdx.x;
    if (!row_live(a.live, img)) return;                // a dead row of the call (efe_rows.mask): workgroup-uniform
    const int Win = a.Win, Hin = a.Hin, Wout = a.Wout, Hout = a.Hout, TY = a.TY;
    const int NR = 2 * TY + 1, WE = (Win + 1) >> 1;
    const int npix_img = Hin * Win;
    const float* src = a.in + (size_t)img * npix_img * IS;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, npix_img * IS * 4, 0x00020000);
    const int c4 = tid & (CQ - 1), ppt = tid / CQ, pstep = 256 / CQ;
    auto fetch = [&](int P, bool in_block) -> float4 {            // pixel P of the image (zeros past its end)
        const unsigned off = (in_block && P < npix_img) ? (unsigned)((P * IS + 4 * c4) * 4) : 0x80000000u;
        const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
        return L == 1 ? make_float4(v.x, v.z, v.y, v.w) : v;
    };
    auto slot_of = [&](int pp, int rs) -> int {                  // pixel pp of a block of rows whose first row has ring slot rs
        const int row = (int)__umulhi((unsigned)pp, a.magicWin), x = pp - row * Win;
        int r = rs + row;
        r = r >= NR ? r - NR : r;
        r = r >= NR ? r - NR : r;
        return (r * Win + ((x & 1) ? WE + (x >> 1) : (x >> 1))) * PS4 + c4;
    };
    // strip 0: input rows 0 .. 2 TY -> ring slots 0 .. NR - 1
    // (NPRO requests in flight together: the whole block for the shapes of BASELINE configs[4] -- batches of 4 were five HBM round trips
    // per image at a third of a layer-2 image's time)
    constexpr int NPRO = L == 1 ? 4 : 20;
    for (int pp0 = ppt; pp0 < NR * Win; pp0 += NPRO * pstep) {
        float4 v[NPRO];
#pragma unroll
        for (int i = 0; i < NPRO; ++i) v[i] = fetch(pp0 + i * pstep, pp0 + i * pstep < NR * Win);
#pragma unroll
        for (int i = 0; i < NPRO; ++i)
            if (pp0 + i * pstep < NR * Win) sx[slot_of(pp0 + i * pstep, 0)] = v[i];
    }
    // this lane's output pixel of a strip
    const int q = w * 32 + j;
    const int yl = (int)__umulhi((unsigned)q, a.magicWout), x = q - yl * Wout;
    float2 w1[9];
    if (L == 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t) w1[t] = reinterpret_cast<const float2*>(a.Wp)[t * 64 + lane];
    }
    const float4* Wl = reinterpret_cast<const float4*>(a.Wp);
    float4 bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(a.bias + 8 * g + 4 * h);
    float* dst = a.out + (size_t)img * Hout * Wout * 32;
    __syncthreads();

    int rs0 = 0;                                       // ring slot of input row 2 y0
    for (int y0 = 0; y0 < Hout; y0 += TY) {
        const int nrows = min(TY, Hout - y0);
        const bool more = y0 + TY < Hout;
        const bool busy = w * 32 < nrows * Wout;        // wave-uniform
        // the next strip's new rows 2 y0 + 2 TY + 1 .. 2 y0 + 4 TY: requested now, written behind the barrier
        float4 pf[NPF];
        int pp0 = ppt; asm volatile("" : "+v"(pp0));    // (laundered: the element offsets are loop invariants hipcc would keep in registers)
        if (more) {
            const int P0 = (2 * y0 + 2 * TY + 1) * Win;
#pragma unroll
            for (int i = 0; i < NPF; ++i) pf[i] = fetch(P0 + pp0 + i * pstep, pp0 + i * pstep < 2 * TY * Win);
        }
        if (busy) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const bool pvx = q < nrows * Wout;
            auto base_of = [&](int t) -> int {          // LDS float4 index of tap t's operand for this lane's pixel
                const int kh = t / 3, kw = t - 3 * kh;
                int r = rs0 + 2 * (pvx ? yl : 0) + kh;
                r = r >= NR ? r - NR : r;
                r = r >= NR ? r - NR : r;
                const int xx = pvx ? x : 0;
                return (r * Win + (kw == 1 ? WE + xx : xx + (kw >> 1))) * PS4;
            };
            if (L == 1) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float2 b = reinterpret_cast<const float2*>(sx + base_of(t))[h];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[t].x, b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[t].y, b.y, acc, 0, 0, 0);
                }
            } else {
                f32x16 (&acc1)[1][1] = reinterpret_cast<f32x16 (&)[1][1]>(acc);
                tap_loop_kc<1, 1, 4>(acc1, 9, Wl, sx, h, [&](int t, int (&bs)[1], int (&sw)[1], int& wt) {
                    wt = t; bs[0] = base_of(t); sw[0] = 0;
                }, PackedWIdx{1, 4, 0});
            }
            if (pvx) {                                   // bias + ReLU; register e holds channel (e & 3) + 8 (e >> 2) + 4 h
                float* op = dst + ((size_t)(y0 + yl) * Wout + x) * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v;
                    v.x = fmaxf(acc[4 * g] + bq[g].x, 0.f); v.y = fmaxf(acc[4 * g + 1] + bq[g].y, 0.f);
                    v.z = fmaxf(acc[4 * g + 2] + bq[g].z, 0.f); v.w = fmaxf(acc[4 * g + 3] + bq[g].w, 0.f);
                    *reinterpret_cast<float4*>(op + 8 * g) = v;
                }
            }
        }
        if (!more) break;
        __syncthreads();                                   // every wave is done reading the rows that are replaced
        {
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                const int pp = pp0 + i * pstep;
                if (pp < 2 * TY * Win) {
                    // new row k (k = 0 .. 2 TY - 1) = image row 2 y0 + 2 TY + 1 + k -> slot (rs0 + 2 TY + 1 + k) mod NR = (rs0 + k) mod NR
                    sx[slot_of(pp, rs0)] = pf[i];
                }
            }
        }
        rs0 += 2 * TY;                                      // row 2 (y0 + TY) = the kept row
        rs0 = rs0 >= NR ? rs0 - NR : rs0;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_conv_e12: layers 1 AND 2 in one kernel -- conv1's output (215 KB per image at 84 x 84: written once and read once by the pair of
// kernels above, 80 % of their HBM traffic) never leaves the chip.  The conv1 rows a strip of conv2 rows needs are the conv2 input ring
// of k_conv_e<2> (same slots, same de-interleaved columns), PRODUCED in place of fetched: a strip is an L1 phase (every wave computes
// 32-pixel tiles of the strip's new conv1 rows, round-robin, and writes bias + ReLU as four float4 per lane into the ring) and the L2
// phase of k_conv_e<2>, two barriers apart.  conv1's operands come straight from the NHWC4 image in global memory (16 B per pixel: a tap
// is two dwords per lane, channels h and 2 + h; a wave's requests cover whole 128-byte lines, the 9 / 4 re-reads hit the L1), requested
// one tile ahead -- the first tile of the NEXT strip before this strip's L2 phase -- so no image ring is needed and two workgroups fit
// a CU.  Every conv1 / conv2 element sees the operations of k_conv_e<1> / <2> in the same order: bit-identical to the two launches.
// ---------------------------------------------------------------------------------------------------------
typedef unsigned u32x2e __attribute__((ext_vector_type(2)));
#ifndef EFE_E12_EARLY
#define EFE_E12_EARLY 0
#endif
#ifndef EFE_E12_PD
#define EFE_E12_PD 4
#endif
constexpr int E12_ROUNDS = 5;
constexpr int E12_PD = EFE_E12_PD;                  // weight fragments of the L2 phase requested this many 8-channel blocks ahead (a tap = 4 blocks = 1024 MFMA cycles)
constexpr bool E12_EARLY = EFE_E12_EARLY != 0;      // request a block's operands before the previous strip's L2 phase instead of behind its barrier
__global__ void __launch_bounds__(256, 2) k_conv_e12(const ConvE12Args a) {
    extern __shared__ __attribute__((aligned(16))) float4 sx[];        // [NR ring rows][W1 slots][9]: conv1 + ReLU, 32 channels + 1 pad quad
    constexpr int PS4 = 9;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int img = blockIdx.x;
    if (!row_live(a.live, img)) return;                // a dead row of the call (efe_rows.mask): workgroup-uniform
    const int W0 = a.W0, W1 = a.W1, W2 = a.W2, H2 = a.H2, TY = a.TY;
    const int NR = 2 * TY + 1, WE = (W1 + 1) >> 1;
    const char* ximg = reinterpret_cast<const char*>(a.in + (size_t)img * a.H0 * W0 * GEN_IMG_LD);
    const unsigned rowb = (unsigned)W0 * GEN_IMG_LD * 4u;           // bytes of an image row
    float2 w1[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w1[t] = reinterpret_cast<const float2*>(a.W1p)[t * 64 + lane];
    float4 bq1[4], bq2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { bq1[g] = *reinterpret_cast<const float4*>(a.b1 + 8 * g + 4 * h); bq2[g] = *reinterpret_cast<const float4*>(a.b2 + 8 * g + 4 * h); }
    const float4* Wl = reinterpret_cast<const float4*>(a.W2p);
    float* dst = a.out + (size_t)img * H2 * W2 * 32;

    // ---- L1: tile T of a block of new conv1 rows = its pixels 32 T .. 32 T + 31 (row-major over the block)
    auto l1_req = [&](float (&v)[18], int T, int first_row, int npx) {
        const int p = T * 32 + j;
        const int rk = (int)__umulhi((unsigned)p, a.magicW1), x = p - rk * W1;
        // image pixel (2 R, 2 x), channels 2 h and 2 h + 1: ONE dwordx2 per tap and lane (the two halves of a wave fetch whole pixels; as two
        // dwords per tap -- channels h and 2 + h, what the MFMA steps consume -- the requests cost 28 cycles each in the texture path and a
        // quarter of the kernel's time).  A lane without a pixel reads the block's first pixel (its tile stores nothing for it).
        // (plain global loads, uniform row base + 32-bit lane offset: this hipcc lowers __builtin_amdgcn_raw_buffer_load_b64 to ONE dword)
        const unsigned off = p < npx ? (unsigned)(2 * rk) * rowb + (unsigned)(2 * x * GEN_IMG_LD + 2 * h) * 4u : (unsigned)(2 * h) * 4u;
        const char* r0 = ximg + (size_t)(2 * first_row) * rowb;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t - 3 * kh;
            const float2 q = *reinterpret_cast<const float2*>(r0 + (size_t)kh * rowb + (off + (unsigned)(kw * GEN_IMG_LD * 4)));
            v[2 * t] = q.x; v[2 * t + 1] = q.y;
        }
    };
    auto l1_tile = [&](const float (&v)[18], int T, int npx, int slot0) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // lane half 0 holds channels (0, 1) of its pixel, half 1 channels (2, 3); the MFMA steps contract (0 | 1) and (2 | 3) -- the
            // order of k_conv_e<1> and k_conv_g: v_permlane32_swap exchanges the upper half of the first register with the lower half of the second
            // (as inline assembly with its own wait states: through __builtin_amdgcn_permlane32_swap this hipcc fed the FIRST result to both MFMAs)
            float c01 = v[2 * t], c23 = v[2 * t + 1];
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c01), "+v"(c23));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[t].x, c01, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[t].y, c23, acc, 0, 0, 0);
        }
        const int p = T * 32 + j;
        if (p < npx) {                                  // bias + ReLU; register e holds channel (e & 3) + 8 (e >> 2) + 4 h = quad 2 (e >> 2) + h
            const int rk = (int)__umulhi((unsigned)p, a.magicW1), x = p - rk * W1;
            int r = slot0 + rk;
            r = r >= NR ? r - NR : r;
            r = r >= NR ? r - NR : r;
            float4* op = sx + (r * W1 + ((x & 1) ? WE + (x >> 1) : (x >> 1))) * PS4 + h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 o;
                o.x = fmaxf(acc[4 * g] + bq1[g].x, 0.f); o.y = fmaxf(acc[4 * g + 1] + bq1[g].y, 0.f);
                o.z = fmaxf(acc[4 * g + 2] + bq1[g].z, 0.f); o.w = fmaxf(acc[4 * g + 3] + bq1[g].w, 0.f);
                op[2 * g] = o;
            }
        }
    };
    // The operands of ALL of a wave's tiles of a block (at most E12_ROUNDS: the ring holds <= 80 KB = 568 pixels = 18 tiles) are
    // requested together, so a block exposes one memory latency instead of one per tile.
    float v[E12_ROUNDS][18];
    auto l1_req_all = [&](int first_row, int nrows) {
        const int npx = nrows * W1;
#pragma unroll
        for (int r = 0; r < E12_ROUNDS; ++r)
            if ((w + 4 * r) * 32 < npx) l1_req(v[r], w + 4 * r, first_row, npx);
    };
    auto l1_phase = [&](int nrows, int slot0) {
        const int npx = nrows * W1;
#pragma unroll
        for (int r = 0; r < E12_ROUNDS; ++r)
            if ((w + 4 * r) * 32 < npx) l1_tile(v[r], w + 4 * r, npx, slot0);
    };
    // strip 0: conv1 rows 0 .. 2 TY -> ring slots 0 .. NR - 1
    l1_req_all(0, NR);
    l1_phase(NR, 0);
    // this lane's output pixel of a strip
    const int q = w * 32 + j;
    const int yl = (int)__umulhi((unsigned)q, a.magicW2), x2 = q - yl * W2;
    __syncthreads();

    int rs0 = 0;                                       // ring slot of conv1 row 2 y0
    for (int y0 = 0; y0 < H2; y0 += TY) {
        const int nrows = min(TY, H2 - y0);
        const bool more = y0 + TY < H2;
        const bool busy = w * 32 < nrows * W2;          // wave-uniform
        // the next strip's new conv1 rows 2 (y0 + TY) + 1 .. : their first tile is requested now, computed behind the barrier
        const int nfirst = 2 * (y0 + TY) + 1, nnew = 2 * min(TY, H2 - y0 - TY);
        if (more && E12_EARLY) l1_req_all(nfirst, nnew);
        if (busy) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const bool pvx = q < nrows * W2;
            auto base_of = [&](int t) -> int {          // LDS float4 index of tap t's operand for this lane's pixel
                const int kh = t / 3, kw = t - 3 * kh;
                int r = rs0 + 2 * (pvx ? yl : 0) + kh;
                r = r >= NR ? r - NR : r;
                r = r >= NR ? r - NR : r;
                const int xx = pvx ? x2 : 0;
                return (r * W1 + (kw == 1 ? WE + xx : xx + (kw >> 1))) * PS4;
            };
            f32x16 (&acc1)[1][1] = reinterpret_cast<f32x16 (&)[1][1]>(acc);
            tap_loop_kc_pd<1, 1, 4, E12_PD>(acc1, 9, Wl, sx, h, [&](int t, int (&bs)[1], int (&sw)[1], int& wt) {
                wt = t; bs[0] = base_of(t); sw[0] = 0;
            }, PackedWIdx{1, 4, 0});
            if (pvx) {
                float* op = dst + ((size_t)(y0 + yl) * W2 + x2) * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v;
                    v.x = fmaxf(acc[4 * g] + bq2[g].x, 0.f); v.y = fmaxf(acc[4 * g + 1] + bq2[g].y, 0.f);
                    v.z = fmaxf(acc[4 * g + 2] + bq2[g].z, 0.f); v.w = fmaxf(acc[4 * g + 3] + bq2[g].w, 0.f);
                    *reinterpret_cast<float4*>(op + 8 * g) = v;
                }
            }
        }
        if (!more) break;
        __syncthreads();                                   // every wave is done reading the rows that are replaced
        if (!E12_EARLY) l1_req_all(nfirst, nnew);
        l1_phase(nnew, rs0);                               // new row k -> slot (rs0 + 2 TY + 1 + k) mod NR = (rs0 + k) mod NR
        rs0 += 2 * TY;                                      // row 2 (y0 + TY) = the kept row
        rs0 = rs0 >= NR ? rs0 - NR : rs0;
        __syncthreads();
    }
}

static int conv_e_ty(const ConvEArgs& a) {
    int ty = 128 / a.Wout;
    if (ty > a.Hout) ty = a.Hout;
    while (ty > 1 && ty * a.Win > 256) --ty;              // the strip's 2 TY new rows must fit the register prefetch (16 float4 per thread at 32 channels)
    return ty;
}
static size_t conv_e_lds(const ConvEArgs& a, int L, int ty) { return (size_t)(2 * ty + 1) * a.Win * (L == 1 ? 1 : 9) * sizeof(float4); }
constexpr size_t CONV_E_MAX_LDS = 96 * 1024;
constexpr size_t CONV_E12_MAX_LDS = 80 * 1024;
int init_generic_enc_kernels() {
    if (hipFuncSetAttribute((const void*)k_conv_e<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONV_E_MAX_LDS) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_conv_e<2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONV_E_MAX_LDS) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_conv_e12, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONV_E12_MAX_LDS) != hipSuccess) return 1;
    return 0;
}
// 0 = launched; 1 = this layer / geometry is outside the kernel's limits (the caller uses k_conv_g)
int launch_conv_e(ConvEArgs a, int layer, hipStream_t st) {
    if (layer != 1 && layer != 2) return 1;
    if (a.Wout < 2 || a.Wout > 128 || a.Win > 255 || a.Hout < 1) return 1;
    a.TY = conv_e_ty(a);
    if (a.TY < 1 || 2 * a.TY * a.Win * (layer == 1 ? 1 : 8) > 256 * (layer == 1 ? 4 : 16)) return 1;
    const size_t lds = conv_e_lds(a, layer, a.TY);
    if (lds > CONV_E_MAX_LDS) return 1;
    a.magicWin = (unsigned)((0x100000000ull + (unsigned)a.Win - 1) / (unsigned)a.Win);
    a.magicWout = (unsigned)((0x100000000ull + (unsigned)a.Wout - 1) / (unsigned)a.Wout);
    if (layer == 1) hipLaunchKernelGGL((k_conv_e<1, 4>), dim3((unsigned)a.n_img), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_conv_e<2, 16>), dim3((unsigned)a.n_img), dim3(256), lds, st, a);
    return 0;
}

// layers 1 + 2 fused.  0 = launched; 1 = outside the kernel's limits (the caller launches the layers one by one)
int launch_conv_e12(ConvE12Args a, hipStream_t st) {
    if (a.W2 < 2 || a.W2 > 128 || a.H2 < 1 || a.W1 > 255 || a.H1 < 2 * a.H2 + 1 || a.W1 < 2 * a.W2 + 1) return 1;
    int ty = 128 / a.W2;
    if (ty > a.H2) ty = a.H2;
    // the conv1 ring of a strip: at most 80 KB, so that two workgroups share a CU
    while (ty > 1 && (size_t)(2 * ty + 1) * a.W1 * 9 * sizeof(float4) > CONV_E12_MAX_LDS) --ty;
    a.TY = ty;
    const size_t lds = (size_t)(2 * ty + 1) * a.W1 * 9 * sizeof(float4);
    if (lds > CONV_E12_MAX_LDS) return 1;
    a.magicW1 = (unsigned)((0x100000000ull + (unsigned)a.W1 - 1) / (unsigned)a.W1);
    a.magicW2 = (unsigned)((0x100000000ull + (unsigned)a.W2 - 1) / (unsigned)a.W2);
    hipLaunchKernelGGL(k_conv_e12, dim3((unsigned)a.n_img), dim3(256), lds, st, a);
    return 0;
}

}  // namespace efe
