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
    if (!row_live(a.live, img)) return;                // a dead row of the call (efe_set_row_mask): workgroup-uniform
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

static int conv_e_ty(const ConvEArgs& a) {
    int ty = 128 / a.Wout;
    if (ty > a.Hout) ty = a.Hout;
    while (ty > 1 && ty * a.Win > 256) --ty;              // the strip's 2 TY new rows must fit the register prefetch (16 float4 per thread at 32 channels)
    return ty;
}
static size_t conv_e_lds(const ConvEArgs& a, int L, int ty) { return (size_t)(2 * ty + 1) * a.Win * (L == 1 ? 1 : 9) * sizeof(float4); }
constexpr size_t CONV_E_MAX_LDS = 96 * 1024;
int init_generic_enc_kernels() {
    if (hipFuncSetAttribute((const void*)k_conv_e<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONV_E_MAX_LDS) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_conv_e<2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONV_E_MAX_LDS) != hipSuccess) return 1;
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

}  // namespace efe
