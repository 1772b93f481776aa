// Decoder convolution kernels of the geometry-generic path (BASELINE configs[4]: 3 x 84 x 84 observations): the
// layers of /root/reference/src/torchmodel.py:120-127 with the layer geometry as run-time arguments, activations NHWC in HBM.
//
//   k_convt_p<MODE, NPF> : ConvTranspose2d(k3, s1, p1) (MODE 1) / ConvTranspose2d(k3, s2, p1, op1) (MODE 2, sub-pixel form, SURVEY
//                          appendix A.1) + ReLU, one workgroup per image walking down full-width strips; pixels are the MFMA rows so
//                          that every store instruction writes whole 128-byte NHWC lines
//   k_dec_bg             : the LAST TWO layers fused, ConvTranspose2d(64, 32, s2) + ReLU -> ConvTranspose2d(32, C, s1) + Sigmoid ->
//                          per-image Bernoulli-entropy / reward sums (+ image store): the design of k_dec_b4 (decoder.hip) with
//                          run-time geometry.  y3 (903 KB per image at 84 x 84, 58 % of the unfused path's HBM traffic) never exists.
//
// The input strip of both kernels is a RING OF PIXELS in LDS: pixel P = row * Win + col of the image lives in slot
// (P + PADT * Win) mod RPa, 64 channels + one float4 of padding per slot (17 float4: a 16-lane group of a ds_read_b128 that reads
// 16 consecutive slots at the same channel quad then touches all 16 bank slots once).  No padding columns: consecutive pixels are
// consecutive slots across row ends, and a lane whose tap falls outside the image reads a ZERO region of 16 slots at the slot
// with its own residue mod 16 -- so every operand read is conflict-free whatever Win is (the [row][col + pad] tiles of the
// first version lost 32 - 46 % of their LDS cycles to the skipped slot at every row end: profiles/r2_v6_ai, SQ_LDS_BANK_CONFLICT).
// The ring size RPa is a multiple of 16, so the wrap keeps the residues too.  Consecutive strips share their halo rows in place:
// only the TH new rows of the next strip are fetched (one contiguous block of global memory), into registers behind the last
// weight-fragment request of the current strip (vmcnt retires in order), and written between two barriers.
#include "mfma_pipe.h"
#include <type_traits>

namespace efe {

typedef unsigned u32x4g __attribute__((ext_vector_type(4)));

__host__ __device__ inline int convt_th(int pixels_per_strip, int Win) { return pixels_per_strip / Win; }

// ---------------------------------------------------------------------------------------------------------
// k_convt_p
// ---------------------------------------------------------------------------------------------------------
template <int MODE, int NPF>
__global__ void __launch_bounds__(256, 3) k_convt_p(const ConvGArgs a) {
    __shared__ int cl_off[128];                      // per strip pixel: byte offset of its (first-parity) output pixel for r0 = 0
    extern __shared__ float4 cl_x[];                 // [RPa ring slots + 16 zero slots][Cin / 4 + 1] float4
    constexpr int PADT = MODE == 1 ? 1 : 0;
    constexpr int NV = MODE == 1 ? 9 : 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int Cin = a.Cin, KC = Cin >> 3, C4 = Cin >> 2, PS4 = C4 + 1;
    const int Win = a.Win, Hin = a.Hin;
    const int ntw = 4 / a.mtiles;
    const int TH = (32 * ntw) / Win;
    const int spi = (Hin + TH - 1) / TH;
    const int img = blockIdx.x;
    if (!row_live(a.live, img)) return;                // a dead row of the call (efe_rows.mask): workgroup-uniform
    const int SPX = TH * Win;                          // pixels of a strip
    const int RP = (TH + PADT + 1) * Win, RPa = (RP + 15) & ~15, ZP = RPa;
    const int npix_img = Hin * Win;
    const float* src = a.in + (size_t)img * npix_img * Cin;
    const int sh4 = 31 - __builtin_clz(C4);            // C4 is a power of two
    const int c4 = tid & (C4 - 1), ppt = tid >> sh4, pstep = 256 >> sh4;
    // a buffer resource over the image: a pixel outside it (row -1, rows past the end) gets an out-of-range offset and the hardware
    // returns zeros
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, npix_img * Cin * 4, 0x00020000);
    auto fetch = [&](int P, bool in_block) -> float4 {
        const bool ok = in_block && P >= 0 && P < npix_img;
        const unsigned off = ok ? (unsigned)((P * Cin + 4 * c4) * 4) : 0x80000000u;
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
    };
    // strip 0: rows -PADT .. TH -> slots 0 .. RP - 1; the zero region
    // (all requests of the block in flight together: batches of 4 were up to four HBM round trips per image with the workgroup idle)
    for (int pp0 = ppt; pp0 < RP; pp0 += 8 * pstep) {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fetch(pp0 + i * pstep - PADT * Win, pp0 + i * pstep < RP);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (pp0 + i * pstep < RP) cl_x[(pp0 + i * pstep) * PS4 + c4] = v[i];
    }
    for (int i = tid; i < 16 * PS4; i += 256) cl_x[ZP * PS4 + i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 128) {
        const int rw = tid / Win, xw = tid - rw * Win;
        cl_off[tid] = rw < TH ? ((MODE == 2 ? 2 * rw : rw) * a.Wout + (MODE == 2 ? 2 * xw : xw)) * a.ldo * 4 : 0x40000000;     // bytes; the sentinel is outside every image
    }
    const int nt = wave % ntw, mt = wave / ntw;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wp), 0, 0x7fffffff, 0x00020000);
    const unsigned wl = (unsigned)lane * 16u;
    // The first weight fragments of a pass are requested BEFORE the previous pass's store epilogue (and, for a strip's first pass,
    // before the barriers in front of it): requested at the top of a pass they exposed an L2 round trip per pass -- two per strip on
    // the stride-2 layers.  fa = fragment set of channel block 0 of the upcoming pass, fb = block 1 (pass 0 keeps two sets).
    constexpr int ptp_[3][9] = {{4, 5, 3, 0, 0, 0, 0, 0, 0}, {7, 8, 6, 1, 2, 0, 0, 0, 0}, {0, 1, 2, 3, 4, 5, 6, 7, 8}};
    float4 fa[9], fb[3];
    auto preload = [&](int P) {        // P = table row of the upcoming pass
        const int n = P == 0 ? 3 : P == 1 ? 6 : 9;
#pragma unroll
        for (int m = 0; m < 9; ++m)
            if (m < n) fa[m] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, wl, (unsigned)(((ptp_[P][m] * a.mtiles + mt) * KC) * 64) * 16u, 0));
        if (P == 0) {
#pragma unroll
            for (int m = 0; m < 3; ++m) fb[m] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, wl, (unsigned)(((ptp_[0][m] * a.mtiles + mt) * KC + 1) * 64) * 16u, 0));
        }
    };
    preload(MODE == 2 ? 0 : 2);
    __syncthreads();

    const int q = nt * 32 + j;
    const bool qv = q < SPX;
    const int qq = qv ? q : 0;
    const int row = qq / Win, x = qq - row * Win;
    const int co = mt * 32 + j;
    const float bias = co < a.Cout ? a.bias[co] : 0.0f;
    // a strip pixel outside the image has a first-parity offset >= the image size: dropped whether or not the scalar parity offset is part of the check
    const int img_floats = a.Hout * a.Wout * a.ldo;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)img * img_floats, 0, img_floats * 4, 0x00020000);
    const int strip_floats = (MODE == 2 ? 2 : 1) * TH * a.Wout * a.ldo;

    int bs = PADT * Win;                               // ring slot of pixel (r0, 0)
    for (int s = 0; s < spi; ++s) {
        const int r0 = s * TH;
        const int nq = min(TH, Hin - r0) * Win;
        const bool more = s + 1 < spi;
        const bool busy = nt * 32 < nq;                // wave-uniform: a short last strip leaves waves without pixels
        // operand views of this strip: view v = pixel (row + dy, x + dx) of the strip; a tap outside the image's columns (or a lane
        // without a pixel) reads the zero region at its own residue
        int vb[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int dy = MODE == 1 ? 1 - v / 3 : (v >> 1), dx = MODE == 1 ? 1 - v % 3 : (v & 1);
            int nat = bs + q + dy * Win + dx;
            if (MODE == 1 && nat < 0) nat += RPa;
            if (nat >= RPa) nat -= RPa;
            const bool ok = qv && x + dx >= 0 && x + dx < Win;
            vb[v] = (ok ? nat : ZP + (nat & 15)) * PS4 + h;
        }
        float4 pf[NPF];
        // the next strip's new rows r0 + TH + 1 .. r0 + 2 TH are one contiguous block of SPX pixels in global memory; they take the
        // slots of the TH oldest rows.  (the element index is laundered per strip: its offsets are loop invariants that hipcc would
        // otherwise keep in registers across the whole strip loop)
        int pp0 = ppt; asm volatile("" : "+v"(pp0));
        auto request_next = [&]() {
            const int P0 = (r0 + TH + 1) * Win;
#pragma unroll
            for (int i = 0; i < NPF; ++i) pf[i] = fetch(P0 + pp0 + i * pstep, pp0 + i * pstep < SPX);
        };
        if (busy) {
            // Stride 2: two passes over the strip, one per output-row parity (3 taps -> parities (0,0) (0,1); 6 taps -> (1,0) (1,1)): 32
            // accumulator registers live instead of 64, so three waves per SIMD fit.  Stride 1: one pass of nine taps (table row 2).
            auto run_pass = [&](auto PC) {
                constexpr int P = decltype(PC)::value;      // table row: 0 / 1 = the two passes of the stride-2 layers, 2 = the stride-1 layer
                constexpr int NMP = P == 0 ? 3 : P == 1 ? 6 : 9, NVP = P == 0 ? 2 : P == 1 ? 4 : 9, NAC = P == 2 ? 1 : 2;
                constexpr int pvw[3][9] = {{0, 0, 1, 0, 0, 0, 0, 0, 0}, {0, 0, 1, 2, 2, 3, 0, 0, 0}, {0, 1, 2, 3, 4, 5, 6, 7, 8}};
                constexpr int ptp[3][9] = {{4, 5, 3, 0, 0, 0, 0, 0, 0}, {7, 8, 6, 1, 2, 0, 0, 0, 0}, {0, 1, 2, 3, 4, 5, 6, 7, 8}};
                constexpr int pac[3][9] = {{0, 1, 1, 0, 0, 0, 0, 0, 0}, {0, 1, 1, 0, 1, 1, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0}};
                f32x16 ac[NAC];
#pragma unroll
                for (int p = 0; p < NAC; ++p)
#pragma unroll
                    for (int e = 0; e < 16; ++e) ac[p][e] = bias;       // a lane owns one output channel: the bias is the accumulator's start value
                // one contraction step: the strip views of block kc (requested a step earlier) against fragment set av; every
                // fragment is re-requested for block kc + AD right behind the MFMAs that consumed it, every view for block kc + 1
                // behind its last reader, so each wait leaves the newer requests in flight (a bulk request per step made hipcc
                // wait for all of them in the middle of the step)
                constexpr int vlast[3][9] = {{1, 2, 0, 0, 0, 0, 0, 0, 0}, {1, 2, 4, 5, 0, 0, 0, 0, 0}, {0, 1, 2, 3, 4, 5, 6, 7, 8}};      // last MFMA group that reads view v
                // (pass 0 keeps two fragment sets, AD = 2 blocks ahead: its three groups are only 768 cycles; pass 1 refills one set, AD = 1)
                constexpr int AD = P == 0 ? 2 : 1;
                auto step = [&](float4 (&av)[NMP], float4 (&bv)[NVP], int kc) {
#pragma unroll
                    for (int m = 0; m < NMP; ++m) {
                        const float4 b = bv[pvw[P][m]];
                        f32x16& c = ac[pac[P][m]];
                        c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, av[m].x, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, av[m].y, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, av[m].z, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, av[m].w, c, 0, 0, 0);
                        if (kc + AD < KC) {
                            const u32x4g v = __builtin_amdgcn_raw_buffer_load_b128(wr, wl, (unsigned)(((ptp[P][m] * a.mtiles + mt) * KC + kc + AD) * 64) * 16u, 0);
                            av[m] = __builtin_bit_cast(float4, v);
                        }
#pragma unroll
                        for (int v = 0; v < NVP; ++v)
                            if (vlast[P][v] == m && kc + 1 < KC) bv[v] = cl_x[vb[v] + 2 * (kc + 1)];      // view v is free: block kc + 1 in place
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                float4 a0[NMP], a1[AD == 2 ? NMP : 1], bv[NVP];
#pragma unroll
                for (int m = 0; m < NMP; ++m) a0[m] = fa[m];                  // requested before the previous pass's epilogue / the strip barriers
                if (AD == 2) {
#pragma unroll
                    for (int m = 0; m < NMP; ++m) a1[m % (AD == 2 ? NMP : 1)] = fb[m % 3];
                }
#pragma unroll
                for (int v = 0; v < NVP; ++v) bv[v] = cl_x[vb[v]];
                for (int kc = 0; kc < KC; kc += 2) {
                    if (P >= 1 && kc + 2 >= KC && more) request_next();     // behind the strip's last fragment request
                    __builtin_amdgcn_sched_barrier(0);
                    step(a0, bv, kc);
                    if constexpr (AD == 2) step(a1, bv, kc + 1); else step(a0, bv, kc + 1);
                }
                preload(P == 0 ? 1 : P == 1 ? 0 : 2);                          // the next pass's first fragments, ahead of this pass's stores
                // epilogue: C/D layout column = lane & 31 (channel), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) (pixel of the tile).
                // Stores go through a buffer resource that covers exactly this image: a strip pixel outside it (a short last strip, the
                // table's sentinel) has an out-of-range offset and is dropped by the hardware -- no branches around the stores.
                if (co < a.Cout) {
                    const unsigned sbase = (unsigned)(s * strip_floats + co) * 4u;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int4 off = *reinterpret_cast<const int4*>(cl_off + nt * 32 + 8 * g4 + 4 * h);
                        const unsigned offs[4] = {(unsigned)off.x, (unsigned)off.y, (unsigned)off.z, (unsigned)off.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const unsigned o = offs[i] + sbase;
#pragma unroll
                            for (int pw = 0; pw < NAC; ++pw) {
                                float v = ac[pw][4 * g4 + i];
                                if (a.relu) v = fmaxf(v, 0.0f);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, o, P == 2 ? 0u : (unsigned)((P * a.Wout + pw) * a.ldo) * 4u, 0);
                            }
                        }
                    }
                }
            };
            if constexpr (MODE == 2) {
                run_pass(std::integral_constant<int, 0>{});
                run_pass(std::integral_constant<int, 1>{});
            } else {
                run_pass(std::integral_constant<int, 2>{});
            }
        } else if (more) {
            request_next();
        }
        if (!more) break;
        __syncthreads();                                   // every wave is done reading the rows that are replaced
        {
            int nb = bs + (TH + 1) * Win;                  // slot of the first new pixel = slot of the oldest row
            if (nb >= RPa) nb -= RPa;
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                const int pp = pp0 + i * pstep;
                int sl = nb + pp;
                if (sl >= RPa) sl -= RPa;
                if (pp < SPX) cl_x[sl * PS4 + c4] = pf[i];
            }
        }
        bs += SPX;
        if (bs >= RPa) bs -= RPa;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_convt_12: ConvTranspose2d(64, 64, k3, s1, p1) + ReLU AND ConvTranspose2d(64, 64, k3, s2, p1, op1) + ReLU in one kernel
// (/root/reference/src/torchmodel.py:120-123): the strips of k_convt_p<1> and k_convt_p<2>, interleaved, with layer 1's output kept
// in a SECOND pixel ring in LDS instead of written to and fetched from HBM (y1: 113 KB per image at base 21).  Round i of an image:
//     A(i)   = layer 1 for its rows i TH .. i TH + TH - 1 (the stride-1 pass of k_convt_p over the x ring), bias + ReLU written into
//              half (i & 1) of the y1 ring (2 TH rows)
//     barrier; the x rows of A(i + 1), requested during A(i), go into the x ring
//     B(i-1) = layer 2 for the input rows (i - 1) TH .. i TH - 1 (the two stride-2 passes over the y1 ring; its halo row i TH is the first
//              row A(i) just wrote), stored to y2
//     barrier
// i.e. two barriers per round and NS + 1 rounds per image, where the two kernels had two per strip each.  A row of y1 below the image
// is never read: the views of B point at the zero region for it.  Every element sees the operations of the two launches in the same
// order (bias as start value, taps and channel blocks in the order of the pass tables): bit-identical results.
// ---------------------------------------------------------------------------------------------------------
template <int NPF>
__global__ void __launch_bounds__(256, 2) k_convt_12(const ConvT12Args a) {
    __shared__ int cl_off[128];                      // per strip pixel: byte offset of its (first-parity) output pixel for r0 = 0
    extern __shared__ float4 cl_x[];                 // [x ring RPa][y1 ring RYa][16 zero slots][1 dummy slot] x 17 float4
    constexpr int KC = 8, C4 = 16, PS4 = 17, Cin = 64, MTL = 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int Win = a.Win, Hin = a.Hin, Wout = 2 * Win;
    const int TH = 64 / Win;
    const int NS = (Hin + TH - 1) / TH;
    const int img = blockIdx.x;
    if (!row_live(a.live, img)) return;                // a dead row of the call (efe_rows.mask): workgroup-uniform
    const int SPX = TH * Win;                          // pixels of a strip
    const int RP = (TH + 2) * Win, RPa = (RP + 15) & ~15;      // x ring: rows r0 - 1 .. r0 + TH
    const int RY = 2 * SPX, YB = RPa, ZP = RPa + ((RY + 15) & ~15);
    const int npix_img = Hin * Win;
    const float* src = a.in + (size_t)img * npix_img * Cin;
    const int c4 = tid & (C4 - 1), ppt = tid >> 4, pstep = 16;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, npix_img * Cin * 4, 0x00020000);
    auto fetch = [&](int P, bool in_block) -> float4 {            // pixel P of the image: zeros above / below it
        const bool ok = in_block && P >= 0 && P < npix_img;
        const unsigned off = ok ? (unsigned)((P * Cin + 4 * c4) * 4) : 0x80000000u;
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
    };
    // round 0's x rows -1 .. TH -> slots 0 .. RP - 1; the zero region
    for (int pp0 = ppt; pp0 < RP; pp0 += 8 * pstep) {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fetch(pp0 + i * pstep - Win, pp0 + i * pstep < RP);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (pp0 + i * pstep < RP) cl_x[(pp0 + i * pstep) * PS4 + c4] = v[i];
    }
    for (int i = tid; i < 16 * PS4; i += 256) cl_x[ZP * PS4 + i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 128) {
        const int rw = tid / Win, xw = tid - rw * Win;
        cl_off[tid] = rw < TH ? (2 * rw * Wout + 2 * xw) * 64 * 4 : 0x40000000;     // bytes; the sentinel is outside every image
    }
    const int nt = wave & 1, mt = wave >> 1;
    // (the wave's feature tile is part of the resource base: every fragment offset below is a compile-time constant -- as scalar
    // expressions of mt hipcc kept ~30 of them in SGPRs and spilled them to VGPR lanes)
    const __amdgpu_buffer_rsrc_t wr1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W1p) + mt * (KC * 256), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W2p) + mt * (KC * 256), 0, 0x7fffffff, 0x00020000);
    const unsigned wl = (unsigned)lane * 16u;
    // first weight fragments of the upcoming pass (see k_convt_p): table row 2 = the stride-1 pass (layer 1), rows 0 / 1 = the stride-2 passes (layer 2)
    constexpr int ptp_[3][9] = {{4, 5, 3, 0, 0, 0, 0, 0, 0}, {7, 8, 6, 1, 2, 0, 0, 0, 0}, {0, 1, 2, 3, 4, 5, 6, 7, 8}};
    float4 fa[9], fb[3];
    auto preload = [&](int P) {
        const int n = P == 0 ? 3 : P == 1 ? 6 : 9;
        const __amdgpu_buffer_rsrc_t wr = P == 2 ? wr1 : wr2;
#pragma unroll
        for (int m = 0; m < 9; ++m)
            if (m < n) fa[m] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, wl, (unsigned)((ptp_[P][m] * MTL * KC) * 64) * 16u, 0));
        if (P == 0) {
#pragma unroll
            for (int m = 0; m < 3; ++m) fb[m] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, wl, (unsigned)((ptp_[0][m] * MTL * KC + 1) * 64) * 16u, 0));
        }
    };
    preload(2);
    __syncthreads();

    const int q = nt * 32 + j;
    const bool qv = q < SPX;
    const int qq = qv ? q : 0;
    const int row = qq / Win, x = qq - row * Win;
    const int co = mt * 32 + j;
    const float bias1 = a.b1[co], bias2 = a.b2[co];
    const int img_floats = 4 * npix_img * 64;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)img * img_floats, 0, img_floats * 4, 0x00020000);
    const int strip_floats = 2 * TH * Wout * 64;
    float* const y1f = reinterpret_cast<float*>(cl_x);
    const int ydummy = (ZP + 16) * (PS4 * 4) + lane;   // float index of this lane's dummy slot (one more pixel slot behind the zero region)

    // one pass over this wave's tile: the contraction of k_convt_p (fragment / view schedule unchanged)
    float4 pf[NPF];
    int pp0 = 0;
    int rA0 = 0; bool moreA = false;
    auto request_next = [&]() {                        // the x rows rA0 + TH + 1 .. rA0 + 2 TH of round i + 1: one contiguous block of SPX pixels
        const int P0 = (rA0 + TH + 1) * Win;
#pragma unroll
        for (int i = 0; i < NPF; ++i) pf[i] = fetch(P0 + pp0 + i * pstep, pp0 + i * pstep < SPX);
    };
    auto run_pass = [&](auto PC, const int (&vb)[9], auto epilogue) {
        constexpr int P = decltype(PC)::value;
        constexpr int NMP = P == 0 ? 3 : P == 1 ? 6 : 9, NVP = P == 0 ? 2 : P == 1 ? 4 : 9, NAC = P == 2 ? 1 : 2;
        constexpr int pvw[3][9] = {{0, 0, 1, 0, 0, 0, 0, 0, 0}, {0, 0, 1, 2, 2, 3, 0, 0, 0}, {0, 1, 2, 3, 4, 5, 6, 7, 8}};
        constexpr int ptp[3][9] = {{4, 5, 3, 0, 0, 0, 0, 0, 0}, {7, 8, 6, 1, 2, 0, 0, 0, 0}, {0, 1, 2, 3, 4, 5, 6, 7, 8}};
        constexpr int pac[3][9] = {{0, 1, 1, 0, 0, 0, 0, 0, 0}, {0, 1, 1, 0, 1, 1, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0}};
        constexpr int vlast[3][9] = {{1, 2, 0, 0, 0, 0, 0, 0, 0}, {1, 2, 4, 5, 0, 0, 0, 0, 0}, {0, 1, 2, 3, 4, 5, 6, 7, 8}};
        constexpr int AD = P == 0 ? 2 : 1;
        const __amdgpu_buffer_rsrc_t wr = P == 2 ? wr1 : wr2;
        const float bias = P == 2 ? bias1 : bias2;
        f32x16 ac[NAC];
#pragma unroll
        for (int p = 0; p < NAC; ++p)
#pragma unroll
            for (int e = 0; e < 16; ++e) ac[p][e] = bias;
        auto step = [&](float4 (&av)[NMP], float4 (&bv)[NVP], int kc) {
#pragma unroll
            for (int m = 0; m < NMP; ++m) {
                const float4 b = bv[pvw[P][m]];
                f32x16& c = ac[pac[P][m]];
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, av[m].x, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, av[m].y, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, av[m].z, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, av[m].w, c, 0, 0, 0);
                if (kc + AD < KC) {
                    const u32x4g v = __builtin_amdgcn_raw_buffer_load_b128(wr, wl, (unsigned)((ptp[P][m] * MTL * KC + kc + AD) * 64) * 16u, 0);
                    av[m] = __builtin_bit_cast(float4, v);
                }
#pragma unroll
                for (int v = 0; v < NVP; ++v)
                    if (vlast[P][v] == m && kc + 1 < KC) bv[v] = cl_x[vb[v] + 2 * (kc + 1)];
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        float4 a0[NMP], a1[AD == 2 ? NMP : 1], bv[NVP];
#pragma unroll
        for (int m = 0; m < NMP; ++m) a0[m] = fa[m];
        if (AD == 2) {
#pragma unroll
            for (int m = 0; m < NMP; ++m) a1[m % (AD == 2 ? NMP : 1)] = fb[m % 3];
        }
#pragma unroll
        for (int v = 0; v < NVP; ++v) bv[v] = cl_x[vb[v]];
        for (int kc = 0; kc < KC; kc += 2) {
            if (P == 2 && kc + 2 >= KC && moreA) request_next();      // behind the layer-1 pass's last fragment request
            __builtin_amdgcn_sched_barrier(0);
            step(a0, bv, kc);
            if constexpr (AD == 2) step(a1, bv, kc + 1); else step(a0, bv, kc + 1);
        }
        epilogue(ac);
    };

    int bs = Win;                                      // x-ring slot of pixel (rA0, 0)
    for (int i = 0; i <= NS; ++i) {
        rA0 = i * TH;
        const bool doA = i < NS, doB = i > 0;
        moreA = i + 1 < NS;
        pp0 = ppt; asm volatile("" : "+v"(pp0));       // (laundered per round: the prefetch offsets are loop invariants hipcc would keep in registers)
        if (doA) {
            const int nq = min(TH, Hin - rA0) * Win;
            if (nt * 32 < nq) {
                int vb[9];
#pragma unroll
                for (int v = 0; v < 9; ++v) {
                    const int dy = 1 - v / 3, dx = 1 - v % 3;
                    int nat = bs + q + dy * Win + dx;
                    if (nat < 0) nat += RPa;
                    if (nat >= RPa) nat -= RPa;
                    const bool ok = qv && x + dx >= 0 && x + dx < Win;
                    vb[v] = (ok ? nat : ZP + (nat & 15)) * PS4 + h;
                }
                const int ybase = (YB + (i & 1) * SPX) * (PS4 * 4) + co;          // float index of (this half's pixel 0, channel co)
                run_pass(std::integral_constant<int, 2>{}, vb, [&](f32x16 (&ac)[1]) {
                    preload(0);                              // layer 2's first fragments, ahead of the ring writes
                    // C/D layout: column = lane & 31 (channel), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) (pixel of the tile)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {            // (a pixel outside the strip goes to a dummy slot: an address select, not a branch per element)
                        const int qa = nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        y1f[qa < nq ? ybase + qa * (PS4 * 4) : ydummy] = fmaxf(ac[0][e], 0.0f);
                    }
                });
            } else {
                if (moreA) request_next();
                preload(0);
            }
        } else {
            preload(0);
        }
        __syncthreads();                                   // A(i)'s rows are in the y1 ring; nobody reads the x ring any more
        if (moreA) {                                       // the x rows of A(i + 1) take the slots of the TH oldest rows
            int nb = bs + (TH + 1) * Win;
            if (nb >= RPa) nb -= RPa;
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                const int pp = pp0 + k * pstep;
                int sl = nb + pp;
                if (sl >= RPa) sl -= RPa;
                if (pp < SPX) cl_x[sl * PS4 + c4] = pf[k];
            }
        }
        bs += SPX;
        if (bs >= RPa) bs -= RPa;
        if (doB) {
            const int s = i - 1, rB0 = s * TH;
            const int nq = min(TH, Hin - rB0) * Win;
            if (nt * 32 < nq) {
                int vb[9];
                const int hb = (s & 1) * SPX;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int dy = v >> 1, dx = v & 1;
                    int nat = hb + q + dy * Win + dx;
                    if (nat >= RY) nat -= RY;
                    const bool ok = qv && x + dx < Win && rB0 + row + dy < Hin;
                    vb[v] = (ok ? YB + nat : ZP + (nat & 15)) * PS4 + h;
                }
#pragma unroll
                for (int v = 4; v < 9; ++v) vb[v] = vb[0];
                auto store = [&](auto PC, auto& ac) {
                    constexpr int P = decltype(PC)::value;
                    const unsigned sbase = (unsigned)(s * strip_floats + co) * 4u;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int4 off = *reinterpret_cast<const int4*>(cl_off + nt * 32 + 8 * g4 + 4 * h);
                        const unsigned offs[4] = {(unsigned)off.x, (unsigned)off.y, (unsigned)off.z, (unsigned)off.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned o = offs[k] + sbase;
#pragma unroll
                            for (int pw = 0; pw < 2; ++pw)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(ac[pw][4 * g4 + k], 0.0f)), yr, o, (unsigned)((P * Wout + pw) * 64) * 4u, 0);
                        }
                    }
                };
                run_pass(std::integral_constant<int, 0>{}, vb, [&](f32x16 (&ac)[2]) { preload(1); store(std::integral_constant<int, 0>{}, ac); });
                run_pass(std::integral_constant<int, 1>{}, vb, [&](f32x16 (&ac)[2]) { preload(2); store(std::integral_constant<int, 1>{}, ac); });
            } else {
                preload(2);
            }
        } else {
            preload(2);
        }
        __syncthreads();                                   // B(i - 1) is done with the half A(i + 1) overwrites; the x ring is complete
    }
}

static size_t convt_p_lds(const ConvGArgs& a) {
    const int ntw = 4 / a.mtiles, TH = (32 * ntw) / a.Win, padt = a.mode == 1 ? 1 : 0;
    const int RPa = ((TH + padt + 1) * a.Win + 15) & ~15;
    return (size_t)(RPa + 16) * (a.Cin / 4 + 1) * sizeof(float4);
}
static int convt_p_npf(const ConvGArgs& a) {
    const int ntw = 4 / a.mtiles, TH = (32 * ntw) / a.Win;
    return (TH * a.Win * (a.Cin / 4) + 255) / 256;
}
constexpr size_t CONVT_P_MAX_LDS = 64 * 1024;
// the LDS-tiled kernel takes the decoder's transposed layers when a full-width strip fits: Cin a power of two >= 16, one or two
// 32-channel output tiles, Win <= 32 * (4 / mtiles), the strip's new rows in <= 8 registers per thread
bool convt_p_ok(const ConvGArgs& a) {
    if (a.mode != 1 && a.mode != 2) return false;
    if ((a.Cin & 15) || (a.Cin & (a.Cin - 1)) || a.Cin > 256 || a.mtiles < 1 || a.mtiles > 2 || a.Win > 32 * (4 / a.mtiles) || a.Win < 2) return false;
    if (convt_p_npf(a) > 8) return false;
    return convt_p_lds(a) <= CONVT_P_MAX_LDS;
}
void launch_convt_p(const ConvGArgs& a, hipStream_t st) {
    const size_t lds = convt_p_lds(a);
    const dim3 grid((unsigned)a.n_img), blk(256);
    const bool small = convt_p_npf(a) <= 4;
    if (a.mode == 1) {
        if (small) hipLaunchKernelGGL((k_convt_p<1, 4>), grid, blk, lds, st, a);
        else hipLaunchKernelGGL((k_convt_p<1, 8>), grid, blk, lds, st, a);
    } else {
        if (small) hipLaunchKernelGGL((k_convt_p<2, 4>), grid, blk, lds, st, a);
        else hipLaunchKernelGGL((k_convt_p<2, 8>), grid, blk, lds, st, a);
    }
}

// layers 1 + 2 fused (64 -> 64 -> 64 channels).  0 = launched; 1 = outside the kernel's limits (the caller launches the layers one by one)
constexpr size_t CONVT_12_MAX_LDS = 80 * 1024;             // two workgroups per CU
static size_t convt_12_lds(const ConvT12Args& a) {
    const int TH = 64 / a.Win, SPX = TH * a.Win;
    const int RPa = ((TH + 2) * a.Win + 15) & ~15, RYa = (2 * SPX + 15) & ~15;
    return (size_t)(RPa + RYa + 16 + 1) * 17 * sizeof(float4);      // rings, zero region, one dummy slot
}
int launch_convt_12(const ConvT12Args& a, hipStream_t st) {
    if (a.Win < 2 || a.Win > 64 || a.Hin < 1) return 1;
    const int TH = 64 / a.Win, npf = (TH * a.Win * 16 + 255) / 256;
    const size_t lds = convt_12_lds(a);
    if (lds > CONVT_12_MAX_LDS || npf > 4) return 1;
    hipLaunchKernelGGL((k_convt_12<4>), dim3((unsigned)a.n_img), dim3(256), lds, st, a);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// k_dec_bg: ConvTranspose2d(64, 32, k3, s2, p1, op1) + ReLU -> ConvTranspose2d(32, C, k3, s1, p1) + Sigmoid -> per-image sums,
// /root/reference/src/torchmodel.py:124-127 with torchutils.py:26-37 and torchmodel.py:210-214,289,292 (C <= 3, any Win <= 64).
//
// INPUT-STATIONARY like k_dec_b4: one workgroup (4 waves) per image, strips of TH = 128 / Win input rows (+ 1 halo row, shared with
// the next strip in place); a wave owns 32 consecutive pixels of the strip (flattened row-major, so a tile may span two rows) and
// ALL FOUR output parities of them.  The nine taps read four shifted views of the input (pixel, right neighbour, pixel below,
// below-right), each B fragment feeds 4 / 2 / 2 / 1 independent accumulator chains; channels are the MFMA rows, so the layer-3
// result of a pixel sits in one lane pair and can be contracted further without leaving the registers.
//
// The 32 -> C conv runs on the accumulators as one more MFMA: T[m][pixel] = sum_co Wt[m][co] relu(y3[co][pixel]) with the rows
// m = (kh, c, kw) -- 27 of the tile's 32 rows for C = 3 -- laid out so that the three kw taps of a (kh, c) group are three
// consecutive registers of ONE lane half (groups 0..4 in lanes 0-31, groups 5..8 in lanes 32-63).  The horizontal part of the
// 3 x 3 sum is then formed in registers (own pixel's two column parities + one wave-wide DPP shift left / right):
//     H[kh][c][row][ox] = sum_kw T[kh, kw, c][row][ox + 1 - kw]
// and only the 3 C H-planes of a y3 row go through LDS (a ring of 2 TH + 2 rows); the gather adds three values per output
// element.  A tile boundary that falls inside an image row splits one horizontal sum between two waves: the boundary lanes
// export their half to a small edge array and the gather adds it for the two output columns concerned.
// ---------------------------------------------------------------------------------------------------------
// sigmoid and the two logarithms of a pixel with the hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp each): the
// libm forms are ~50 VALU instructions per output element, and this geometry has C = 3 outputs per pixel (on the dSprites kernels the
// accurate forms stay: parity pinned, one output per pixel).  Error: |x| 2^-24 relative on e^-x, i.e. < 5e-7 absolute on the sigmoid
// for |x| < 30, and < 2 ulp on the logarithms -- inside the fp32 tolerances of tests/test_generic_geometry.py, which compare against
// the libm-evaluated oracle.

// Lane predicates that do not change from strip to strip (is this lane's pixel inside the strip / at an image edge / at a tile edge,
// does this output element exist, does a tile boundary split its horizontal sum) are kept as BIT MASKS IN VGPRs (all ones / zero,
// made opaque to the optimiser once) and applied with v_and / v_bfi: as conditions hipcc hoists each of them out of the strip loop
// as a 64-bit SGPR mask -- about 25 of them, 47 SGPRs spilled to VGPR lanes and read back with v_readlane inside the MFMA phase.
// Everything else that used to be a predicate is an addressing rule instead:
//   * the next strip's rows are fetched as a fixed 128 pixels (8 float4 per thread): pixels past the strip's block land in ring
//     slots nobody reads (the ring has 128 + Win slots, rounded up to 16), pixels past the image are zeros of the buffer resource;
//   * an H-ring row above the image is a zeroed ring row (the two rows in front of strip 0), so the gather adds it unconditionally;
//   * an output element whose horizontal sum is not split reads edge entry 4 of its row group, which always holds zero (no boundary
//     lane ever exports to it: it would be the left neighbour of tile 0);
//   * a lane without a pixel (the last lanes of the last tile when Win does not divide 128) writes its H values to a dummy slot.
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int bsel(int m, int a, int b) { return (a & m) | (b & ~m); }      // v_bfi_b32
__device__ __forceinline__ int wrap1(int x, int n) { return (int)min((unsigned)x, (unsigned)(x - n)); }      // x in [0, 2n) -> x mod n

template <int C>
__global__ void __launch_bounds__(256, 2) k_dec_bg(const DecBGArgs a) {
    extern __shared__ float4 sm[];                    // [RPa ring slots + 16 zero slots][17], then H planes, edge values, dummy slots
    __shared__ float sred[4];
    __shared__ float4 sb3[8];
    constexpr int PS4 = 17, Cin = 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    constexpr int NG = 3 * C;
    const int Win = a.Win, Hin = a.Hin, TH = a.TH;
    const int Wout = 2 * Win, Hout = 2 * Hin;
    const int SPX = TH * Win, RP = (TH + 1) * Win, RPa = a.RPa, ZP = RPa;
    const int RING = 2 * TH + 2;
    float* sH = reinterpret_cast<float*>(sm + (size_t)(RPa + 16) * PS4);      // [RING][NG][Wout]
    float* sE = sH + RING * NG * Wout;                                         // [RING][NG][2 sides][4 tiles][2]
    float* sD = sE + RING * NG * 16;                                           // [64 lanes][2]: where lanes without a pixel / a row group write
    const int img = blockIdx.x;
    if (!row_live(a.live, img)) return;                // a dead row of the call (efe_rows.mask): workgroup-uniform

    const int mg = a.m0 + img;
    const int g = mg / a.rows_per_group;
    const int r = mg - g * a.rows_per_group;
    int gt, gp, gs;
    group_decode(a.gm, g, gt, gp, gs);
    const int mode = (gp == 0 && a.reward0) ? 1 : 0;
    const int slot = (gp == 0 && a.store0) ? gt * a.gm.S + gs : -1;
    float* po = (slot >= 0) ? a.po + ((size_t)slot * a.rows_per_group + r) * ((size_t)Hout * Wout * GEN_IMG_LD) : nullptr;

    const int npix_img = Hin * Win;
    const float* src = a.y2 + (size_t)img * npix_img * Cin;
    // a buffer resource over exactly this image: a pixel past its end has an out-of-range offset and the hardware returns zeros
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, npix_img * Cin * 4, 0x00020000);
    const int c4 = tid & 15, ppt = tid >> 4;           // this thread's channel quad and first pixel of a 16-pixel step: byte tid * 16 of a pixel block
    {   // strip 0: rows 0 .. TH -> slots 0 .. RP - 1 (RP <= 192 pixels = 12 float4 per thread, all requested before the first is written:
        // one HBM round trip per image instead of three)
        float4 v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)(tid * 16 + i * 4096), 0, 0));
#pragma unroll
        for (int i = 0; i < 12; ++i)
            if (ppt + 16 * i < RP) sm[(ppt + 16 * i) * PS4 + c4] = v[i];
    }
    for (int i = tid; i < 16 * PS4; i += 256) sm[ZP * PS4 + i] = make_float4(0.f, 0.f, 0.f, 0.f);
    {   // H ring, edge array and dummy slots start as zeros (the two ring rows in front of strip 0 and edge entry 4 stay zero as long as they are read)
        float4* z = reinterpret_cast<float4*>(sH);
        const int nz = (RING * NG * (Wout + 16) + 128) >> 2;
        for (int i = tid; i < nz; i += 256) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 8) sb3[tid] = reinterpret_cast<const float4*>(a.b3)[tid];

    // A fragments of the tap contraction: row m of the 32 x 32 tile <-> (lane half hm, register u): hm = (m >> 2) & 1,
    // u = (m & 3) + 4 (m >> 3) (the C/D layout: register e' of lane half h holds row (e' & 3) + 8 (e' >> 2) + 4 h); group t = u / 3,
    // kw = u % 3, G = hm ? 5 + t : t = kh * C + c; k = channel 8 (e >> 2) + 4 (lane >> 5) + (e & 3) of MFMA e
    float aw[16];
    {
        const int m = j, hm = (m >> 2) & 1, u = (m & 3) + 4 * (m >> 3);
        const int t = u / 3, kw = u - 3 * t, G = hm ? 5 + t : t;
        const bool mv = u < 15 && G < NG;
        const int kh = mv ? G / C : 0, c = mv ? G - kh * C : 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ci = 8 * (e >> 2) + 4 * h + (e & 3);
            aw[e] = mv ? a.w4[((kh * 3 + kw) * 32 + ci) * 4 + c] : 0.0f;
        }
    }
    // this lane's pixel of a strip
    const int q = w * 32 + j;
    const bool qv = q < SPX;
    const int lrow = (qv ? q : 0) / Win, ix = (qv ? q : 0) - lrow * Win;
    const bool first_col = ix == 0, last_col = ix == Win - 1;
    // strip-invariant lane masks (see the note above the kernel)
    const int m_q = opaque(qv ? -1 : 0);                                  // the lane has a pixel
    const int m_qr = opaque((qv && !last_col) ? -1 : 0);                  // ... and that pixel has a right neighbour
    const int m_l = opaque((j == 0 || first_col) ? 0 : -1);               // the left neighbour's value comes from the DPP shift
    const int m_r = opaque((j == 31 || last_col) ? 0 : -1);
    const int m_q0 = opaque((qv && h == 0) ? -1 : 0), m_q1 = opaque((qv && h == 1) ? -1 : 0);      // ... in lane half 0 / 1
    const int m_j31 = opaque(j == 31 ? -1 : 0);
    const int wmask = w > 0 ? -1 : 0;                                     // (uniform) tile 0 has no left neighbour tile: its export slot is the zero entry
    const int lrow2 = 2 * lrow;
    const int hp_lane = (2 * ix + (h ? 5 * Wout : 0)) * 4;                // byte offset of this lane's H values inside a ring row: column 2 ix of row group h ? 5 : 0
    const int ep_lane = (w + (h ? 5 * 8 : 0) + (j == 0 ? 4 : 0)) * 8;     // a tile's lane 31 exports to entry w, its lane 0 to entry 4 + w
    const int dummy = (int)(size_t)sD + lane * 8;                         // LDS byte address

    const float4* W3 = reinterpret_cast<const float4*>(a.w3);
    const __amdgpu_buffer_rsrc_t wr = wrsrc(W3);
    const unsigned ln = (unsigned)lane * 16u;
    auto wf = [&](int tap, int kc) -> float4 { return wfrag(wr, ln, (size_t)(tap * 8 + kc) * 64); };
    const float D1 = 1.00001f, D0 = 0.00001f;
    float part = 0.f;
    const int NS = (Hin + TH - 1) / TH;
    int bs = 0;                                        // ring slot of pixel (r0, 0)
    int hb = 0;                                        // H-ring slot of y3 row 2 r0
    // ---- gather of the output elements of a strip.  A thread owns one COLUMN PAIR (2 xp, 2 xp + 1) of one output row -- the H pair one
    // lane of the tap phase wrote -- so the 2 TH rows of a strip with a successor are one pass of <= 256 threads, every LDS read is a
    // ds_read_b64 and the sums are packed adds.  Pieces:
    //   g_load : the 3 C H pairs and 3 C edge pairs of the thread's element            g_add : v[c] = b4[c] + sum_kh (H[kh][c] + E[kh][c])
    //   g_sig  : sigmoid       g_term : entropy / reward term into the thread's partial sum (image-uniform branch)       g_store : image store
    // All of them run inside the NEXT strip's contraction, between MFMA groups (the H rows they read are rewritten behind that strip's
    // barrier A at the earliest); the last strip of an image is gathered in full behind its barrier B (g_last).
    // Edge entries are PAIRS (to the even column, to the odd column): a tile's lane 31 exports (x, 0) to entry w -- read by the next tile's
    // first lane, whose even column misses its left neighbour's kw = 2 value -- and its lane 0 exports (0, x) to entry 4 + w.  An element
    // whose sums are not split reads entry 4, which always holds (0, 0) (tile 0 exports zero there).
    f32x2 gv[C]; int goh, gxp, gokm;
    const int f2 = tid;                                   // pair index inside the strip's 2 TH output rows
    int forow = (int)__umulhi((unsigned)f2, a.magicWin);
    const int fxp0 = f2 - forow * Win;
    const int m_f = opaque(f2 < 2 * TH * Win ? -1 : 0);
    if (f2 >= 2 * TH * Win) forow = 0;                    // (a thread without an element reads the strip's first rows and is discarded)
    const int fxp8 = opaque((m_f ? fxp0 : 0) * 8);        // byte offset of the pair inside an H row
    int feo8[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        int lr_ = (forow - kh) >> 1;                      // input row of the source y3 row, relative to the strip (-1 = the previous strip's last row)
        if (lr_ < 0) lr_ += TH;
        const int qq = lr_ * Win + (m_f ? fxp0 : 0);
        const bool needL = (qq & 31) == 0 && fxp0 > 0;
        const bool needR = (qq & 31) == 31 && fxp0 < Win - 1;
        feo8[kh] = opaque(((needL ? (qq >> 5) - 1 : needR ? 4 + (qq >> 5) + 1 : 4) + kh * C * 8) * 8);
    }
    const int rowH = NG * Wout * 4, rowE = NG * 8 * 8;      // bytes of a ring row of the H planes / of the edge array
    const int sHb = (int)(size_t)sH, sEb = (int)(size_t)sE;
    auto lds_f2 = [](int byte_addr) -> f32x2 { return *reinterpret_cast<const f32x2 __attribute__((address_space(3)))*>((size_t)byte_addr); };
    f32x2 gh[C], ge[C];
    int g_hbm;
    auto g_begin = [&](int r0g, int hbg) {
        const int oh = 2 * r0g - 1 + forow;
        goh = oh; gxp = fxp8 >> 3;
        gokm = m_f & ~(oh >> 31);                          // (oh < Hout for every strip with a successor)
        g_hbm = hbg - 2; g_hbm = g_hbm < 0 ? g_hbm + RING : g_hbm;          // uniform
#pragma unroll
        for (int c = 0; c < C; ++c) { gv[c].x = a.b4[c]; gv[c].y = a.b4[c]; }
    };
    auto g_load = [&](auto KH) {           // tap row kh: its C H pairs and C edge pairs
        constexpr int kh = decltype(KH)::value;
        // y3 source row = 2 r0 + (forow - kh), ring slot hb + (forow - kh) = (hb - 2) + (2 - kh) + forow, in [0, 2 RING); a source row above
        // the image is one of the two zeroed ring rows in front of strip 0
        const int hs = wrap1(g_hbm + (2 - kh) + forow, RING);
        const int hq = hs * rowH + (sHb + kh * C * Wout * 4) + fxp8;
        const int eq = hs * rowE + sEb + feo8[kh];
#pragma unroll
        for (int c = 0; c < C; ++c) { gh[c] = lds_f2(hq + c * Wout * 4); ge[c] = lds_f2(eq + c * 64); }
    };
    auto g_add = [&]() {                   // in the fixed order b4 + (H0 + E0) + (H1 + E1) + (H2 + E2), one tap row per call
#pragma unroll
        for (int c = 0; c < C; ++c) { gv[c] += gh[c]; gv[c] += ge[c]; asm volatile("" : "+v"(gv[c])); }
    };
    // (the empty asm statements pin a piece where it is written: its results have no reader until after the strip barrier, and LLVM
    // otherwise sinks the whole computation there -- out of the MFMA groups)
    auto g_sig = [&]() {
#pragma unroll
        for (int c = 0; c < C; ++c) { gv[c].x = hw_sigmoid(gv[c].x); gv[c].y = hw_sigmoid(gv[c].y); asm volatile("" : "+v"(gv[c])); }
    };
    auto g_term = [&]() {                   // (every discarded thread's terms are finite: its sums come from valid LDS slots)
        f32x2 t; t.x = 0.f; t.y = 0.f;
        if (mode == 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const f32x2 pr = gv[c];
                f32x2 l1, l0;
                l1.x = hw_log(D1 - pr.x); l1.y = hw_log(D1 - pr.y); l0.x = hw_log(D0 + pr.x); l0.y = hw_log(D0 + pr.y);
                t += (pr - 1.0f) * l1 - pr * l0;          // -(1 - p) ln((1e-5 + 1) - p) - p ln(1e-5 + p)
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                t.x += reward_term(gv[c].x, goh, 2 * gxp, Hout, Wout, a.reward_intent);
                t.y += reward_term(gv[c].y, goh, 2 * gxp + 1, Hout, Wout, a.reward_intent);
            }
        }
        part += __builtin_bit_cast(float, __builtin_bit_cast(int, t.x + t.y) & gokm);
        asm volatile("" : "+v"(part));
    };
    auto g_store = [&]() {
        if (po && gokm) {
            float4* d = reinterpret_cast<float4*>(po + ((size_t)goh * Wout + 2 * gxp) * GEN_IMG_LD);
            d[0] = make_float4(gv[0].x, C > 1 ? gv[C > 1 ? 1 : 0].x : 0.f, C > 2 ? gv[C > 2 ? 2 : 0].x : 0.f, 0.f);
            d[1] = make_float4(gv[0].y, C > 1 ? gv[C > 1 ? 1 : 0].y : 0.f, C > 2 ? gv[C > 2 ? 2 : 0].y : 0.f, 0.f);
        }
    };
    // the last strip's rows 2 r0 - 1 .. 2 r0 + 2 TH - 1 (it has no successor that would take its last row): pair f of the pass, general form
    // (image edges, the strip may be short)
    auto g_last = [&](int fp, int r0g, int nrows) {
        const int f = fp + tid;
        const int orow = (int)__umulhi((unsigned)f, a.magicWin);
        const int xp = f - orow * Win, oh = 2 * r0g - 1 + orow;
        goh = oh; gxp = xp;
        gokm = (f < nrows * Win && oh >= 0 && oh < Hout) ? -1 : 0;
        const int ohc = min(max(oh, 0), Hout - 1);          // (a thread without an output element reads valid slots and is discarded)
        const int xpc = gokm ? xp : 0;
#pragma unroll
        for (int c = 0; c < C; ++c) { gv[c].x = a.b4[c]; gv[c].y = a.b4[c]; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int tr = ohc + 1 - kh;            // y3 source row of tap row kh
            const bool rv = tr >= 0 && tr < Hout;
            int hs = hb + ((rv ? tr : ohc) - 2 * r0g);   // (an out-of-image source row reads a valid slot and adds zero)
            if (hs < 0) hs += RING;
            if (hs >= RING) hs -= RING;
            int lr_ = (tr >> 1) - r0g;
            if (lr_ < 0) lr_ += TH;
            const int qq = lr_ * Win + xpc;
            const bool needL = (qq & 31) == 0 && xpc > 0;
            const bool needR = (qq & 31) == 31 && xpc < Win - 1;
            const int hq = hs * rowH + (sHb + kh * C * Wout * 4) + xpc * 8;
            const int eq = hs * rowE + sEb + ((needL ? (qq >> 5) - 1 : needR ? 4 + (qq >> 5) + 1 : 4) + kh * C * 8) * 8;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const f32x2 hv = lds_f2(hq + c * Wout * 4), ev = lds_f2(eq + c * 64);
                gv[c].x += rv ? hv.x : 0.0f; gv[c].y += rv ? hv.y : 0.0f;
                gv[c].x += rv ? ev.x : 0.0f; gv[c].y += rv ? ev.y : 0.0f;
            }
        }
    };
    gokm = 0; goh = 0; gxp = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) { gv[c].x = 0.f; gv[c].y = 0.f; }
    float4 pf[8];
    float4 a0 = wf(4, 0), a1 = wf(5, 0), a2 = wf(7, 0), a3 = wf(8, 0);      // a strip's first weight fragments are in flight across the barrier in front of it
    __syncthreads();

    // ---- horizontal pre-sum of row group t of lane half h (G = h ? 5 + t : t): ps_calc forms the lane's H pair and its boundary
    // export in registers (DPP shifts + adds), ps_store writes them; hpa / epa = LDS byte addresses of the lane's H pair / edge entries
    // in the ring row of (lrow, ph)
    f32x2 eo[5]; float xe[5];
    auto ps_calc = [&](auto TC, const f32x16& T0, const f32x16& T1) {
        constexpr int t = decltype(TC)::value;
        constexpr bool v0 = t < NG, v1 = 5 + t < NG;         // is G a row group at all, per lane half
        if constexpr (v0 || v1) {
            // out column 2 ix     takes kw = 0 from x = 2 ix + 1, kw = 1 from x = 2 ix, kw = 2 from x = 2 ix - 1 (left neighbour's odd column)
            // out column 2 ix + 1 takes kw = 0 from x = 2 ix + 2 (right neighbour's even column), kw = 1 from x = 2 ix + 1, kw = 2 from x = 2 ix
            float l = wave_shr1(T1[3 * t + 2]), rr = wave_shl1(T0[3 * t]);
            l = __builtin_bit_cast(float, __builtin_bit_cast(int, l) & m_l);       // tile boundary (the neighbour is another wave's lane: edge array) or image edge
            rr = __builtin_bit_cast(float, __builtin_bit_cast(int, rr) & m_r);
            eo[t].x = (T1[3 * t] + T0[3 * t + 1]) + l;
            eo[t].y = (rr + T1[3 * t + 1]) + T0[3 * t + 2];
            // exports of the tile's boundary lanes: lane 31's kw = 2 value is for output column 2 (ix + 1) of the next tile's first lane, lane
            // 0's kw = 0 value for column 2 (ix - 1) + 1 of the previous tile's last lane (tile 0 exports zero: its entry is the 'no split' one)
            // (the elements are copied to scalars first: __builtin_bit_cast on a vector-element lvalue reads element 0)
            const float e31 = T1[3 * t + 2], e00 = T0[3 * t];
            xe[t] = __builtin_bit_cast(float, bsel(m_j31, __builtin_bit_cast(int, e31), __builtin_bit_cast(int, e00) & wmask));
            asm volatile("" : "+v"(eo[t].x), "+v"(eo[t].y), "+v"(xe[t]));
        }
    };
    auto ps_store = [&](auto TC, int hpa, int epa) {
        constexpr int t = decltype(TC)::value;
        constexpr bool v0 = t < NG, v1 = 5 + t < NG;
        if constexpr (v0 || v1) {
            // a lane without a pixel, and the lane half whose G is not a row group, write to their dummy slot
            const int mv = (v0 && v1) ? m_q : v0 ? m_q0 : m_q1;
            const int ha = bsel(mv, hpa + t * Wout * 4, dummy), ea = bsel(mv, epa + t * 64, dummy);
            *reinterpret_cast<f32x2 __attribute__((address_space(3)))*>((size_t)ha) = eo[t];
            if (j == 31 || j == 0) {
                f32x2 x2;       // lane 31: (x, 0) for the next tile's even column; lane 0: (0, x) for the previous tile's odd column
                x2.x = __builtin_bit_cast(float, __builtin_bit_cast(int, xe[t]) & m_j31); x2.y = __builtin_bit_cast(float, __builtin_bit_cast(int, xe[t]) & ~m_j31);
                *reinterpret_cast<f32x2 __attribute__((address_space(3)))*>((size_t)ea) = x2;
            }
        }
    };
    auto ring_addr = [&](int ph, int& hpa, int& epa) {      // y3 row 2 (r0 + lrow) + ph of this lane -> H-ring slot -> byte addresses
        const int hs = wrap1(hb + ph + lrow2, RING);         // hb + 1 + 2 (TH - 1) < 2 RING
        hpa = hs * rowH + sHb + hp_lane;
        epa = hs * rowE + sEb + ep_lane;
    };

    // One strip:   [barrier B]  g_load (previous strip's sums)  |  contraction: view A (+ g_add / g_sig / g_term pieces), view B, tap contraction
    // of output rows 2 y, views C + D (+ the next strip's rows requested)  [barrier A: every wave is done reading the input ring]  tap contraction
    // of rows 2 y + 1 with, between its MFMA groups, the pre-sums and H stores of rows 2 y, the ring writes of the new rows and the previous
    // strip's image stores  |  pre-sums of rows 2 y + 1  [barrier B: H planes and input ring complete].  Beside a co-resident wave that streams
    // MFMAs an instruction issued OUTSIDE the wave's own MFMA groups costs ~19 cycles: only g_load's address arithmetic
    // and LDS requests, the last pre-sums and the loop bookkeeping are left there (the version with every H store, the gather sums and the ring
    // writes between two barriers had ~470 such instructions per strip, this one ~230).
    for (int s = 0; s < NS; ++s) {
        const int r0 = s * TH;
        const bool more = s + 1 < NS;
        // the four shifted views: LDS float4 index of (slot, quad h)
        int pA, pB, pC, pD;
        {
            const int nA = wrap1(bs + q, RPa), nB = wrap1(nA + 1, RPa), nC = wrap1(nA + Win, RPa), nD = wrap1(nC + 1, RPa);
            pA = bsel(m_q, nA, ZP + (nA & 15)) * PS4 + h;
            pB = bsel(m_qr, nB, ZP + (nB & 15)) * PS4 + h;
            pC = bsel(m_q, nC, ZP + (nC & 15)) * PS4 + h;
            pD = bsel(m_qr, nD, ZP + (nD & 15)) * PS4 + h;
        }
        int hbp = hb - 2 * TH; hbp = hbp < 0 ? hbp + RING : hbp;       // the previous strip's output rows 2 (r0 - TH) - 1 .. 2 r0 - 2 are complete
        f32x16 bbv;                                          // register e holds channel (e & 3) + 8 (e >> 2) + 4 h: the bias is the chains' start value
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 bb = sb3[2 * g4 + h];
            bbv[4 * g4] = bb.x; bbv[4 * g4 + 1] = bb.y; bbv[4 * g4 + 2] = bb.z; bbv[4 * g4 + 3] = bb.w;
        }
        f32x16 acc[4];
        // ---- contraction, software-pipelined one chunk ahead: view A (4 chains: taps (1,1) (1,2) (2,1) (2,2) of parities 0..3), view B
        // (taps (1,0) (2,0) of parities 1, 3), views C + D fused (taps (0,1) (0,2) of parities 2, 3; tap (0,0) of parity 3).  Parities 0
        // and 1 (output rows 2 y) are complete after view B.
        f32x16 T0, T1, U0, U1;
        {
            float4 b = sm[pA];
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const float4 c0 = a0, c1 = a1, c2 = a2, c3 = a3, cb = b;
                if (kc < 7) {
                    a0 = wf(4, kc + 1); a1 = wf(5, kc + 1); a2 = wf(7, kc + 1); a3 = wf(8, kc + 1);
                    b = sm[pA + 2 * (kc + 1)];
                } else {
                    a0 = wf(3, 0); a1 = wf(6, 0);
                    b = sm[pB];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kc == 0) { MFMA4I(acc[0], bbv, c0, cb) MFMA4I(acc[1], bbv, c1, cb) MFMA4I(acc[2], bbv, c2, cb) MFMA4I(acc[3], bbv, c3, cb) }
                else { MFMA4(acc[0], c0, cb) MFMA4(acc[1], c1, cb) MFMA4(acc[2], c2, cb) MFMA4(acc[3], c3, cb) }
                // the previous strip's gather, a piece per channel-block step
                if (s > 0) {
                    if (kc == 0) { g_begin(r0 - TH, hbp); g_load(std::integral_constant<int, 0>{}); }
                    if (kc == 1) { g_add(); g_load(std::integral_constant<int, 1>{}); }
                    if (kc == 2) { g_add(); g_load(std::integral_constant<int, 2>{}); }
                    if (kc == 3) g_add();
                    if (kc == 4) g_sig();
                    if (kc == 5) g_term();
                }
            }
            float4 bd;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const float4 c0 = a0, c1 = a1, cb = b;
                if (kc < 7) {
                    a0 = wf(3, kc + 1); a1 = wf(6, kc + 1);
                    b = sm[pB + 2 * (kc + 1)];
                } else {
                    a0 = wf(1, 0); a1 = wf(2, 0); a2 = wf(0, 0);
                    b = sm[pC];
                    bd = sm[pD];
                }
                __builtin_amdgcn_sched_barrier(0);
                MFMA4(acc[1], c0, cb) MFMA4(acc[3], c1, cb)
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- ReLU + tap contraction of output rows 2 y: T = Wt x relu(acc) (two column parities)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[0][e] = relu_bits(acc[0][e]); acc[1][e] = relu_bits(acc[1][e]); T0[e] = 0.f; T1[e] = 0.f; }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                T0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[e], acc[0][e], T0, 0, 0, 0);
                T1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[e], acc[1][e], T1, 0, 0, 0);
            }
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const float4 c0 = a0, c1 = a1, c2 = a2, cb = b, cd = bd;
                if (kc < 7) {
                    a0 = wf(1, kc + 1); a1 = wf(2, kc + 1); a2 = wf(0, kc + 1);
                    b = sm[pC + 2 * (kc + 1)];
                    bd = sm[pD + 2 * (kc + 1)];
                }
                if (kc == 6 && more) {       // the next strip's new rows (128 pixels from the first new one), behind the strip's last weight-fragment request
                    const unsigned P0b = (unsigned)((r0 + TH + 1) * Win) * (Cin * 4);
#pragma unroll
                    for (int i = 0; i < 8; ++i) pf[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)(tid * 16) + (P0b + i * 4096u), 0, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
                MFMA4(acc[3], c1, cb) MFMA4(acc[2], c0, cb) MFMA4(acc[3], c2, cd)
                // rows 2 y: pre-sums in registers (their LDS stores wait for barrier A: other waves still read the H rows they replace)
                if (kc == 1) ps_calc(std::integral_constant<int, 0>{}, T0, T1);
                if (kc == 2) ps_calc(std::integral_constant<int, 1>{}, T0, T1);
                if (kc == 3) ps_calc(std::integral_constant<int, 2>{}, T0, T1);
                if (kc == 4) ps_calc(std::integral_constant<int, 3>{}, T0, T1);
                if (kc == 5) ps_calc(std::integral_constant<int, 4>{}, T0, T1);
            }
        }
        __syncthreads();                                   // barrier A: no wave reads this strip's rows of the input ring any more
        // ---- output rows 2 y + 1 (parities 2, 3): tap contraction in four groups of eight MFMAs, the other work between them
        int hpa, epa;
        ring_addr(0, hpa, epa);
        int nb = bs + RP; nb = nb >= RPa ? nb - RPa : nb;       // uniform: slot of pixel (r0 + TH + 1, 0)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[2][e] = relu_bits(acc[2][e]); acc[3][e] = relu_bits(acc[3][e]); U0[e] = 0.f; U1[e] = 0.f; }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
            for (int e = 4 * g4; e < 4 * g4 + 4; ++e) {
                U0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[e], acc[2][e], U0, 0, 0, 0);
                U1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[e], acc[3][e], U1, 0, 0, 0);
            }
            if (g4 == 0) { ps_store(std::integral_constant<int, 0>{}, hpa, epa); ps_store(std::integral_constant<int, 1>{}, hpa, epa); }
            if (g4 == 1) { ps_store(std::integral_constant<int, 2>{}, hpa, epa); ps_store(std::integral_constant<int, 3>{}, hpa, epa); }
            if (g4 == 2) {
                ps_store(std::integral_constant<int, 4>{}, hpa, epa);
                if (more) {
                    // the new rows take the slots behind the halo row; the pixels past the block's TH rows go to slots that no strip reads
                    // before they are rewritten (the ring has >= 128 + Win slots)
#pragma unroll
                    for (int i = 0; i < 4; ++i) sm[wrap1(nb + ppt + 16 * i, RPa) * PS4 + c4] = pf[i];
                }
            }
            if (g4 == 3) {
                if (more) {
#pragma unroll
                    for (int i = 4; i < 8; ++i) sm[wrap1(nb + ppt + 16 * i, RPa) * PS4 + c4] = pf[i];
                }
                g_store();                                 // the previous strip's pixels (sigmoid piece: view A section)
                if (more) { a0 = wf(4, 0); a1 = wf(5, 0); a2 = wf(7, 0); a3 = wf(8, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        ring_addr(1, hpa, epa);
        ps_calc(std::integral_constant<int, 0>{}, U0, U1); ps_store(std::integral_constant<int, 0>{}, hpa, epa);
        ps_calc(std::integral_constant<int, 1>{}, U0, U1); ps_store(std::integral_constant<int, 1>{}, hpa, epa);
        ps_calc(std::integral_constant<int, 2>{}, U0, U1); ps_store(std::integral_constant<int, 2>{}, hpa, epa);
        ps_calc(std::integral_constant<int, 3>{}, U0, U1); ps_store(std::integral_constant<int, 3>{}, hpa, epa);
        ps_calc(std::integral_constant<int, 4>{}, U0, U1); ps_store(std::integral_constant<int, 4>{}, hpa, epa);
        __syncthreads();                                   // barrier B: this strip's H planes and the next strip's rows are in LDS
        if (!more) {
            // the last strip has no successor: its rows (and the image's last row) are gathered in full now
            const int nrows = 2 * TH + 1;
            for (int fp = 0; fp < nrows * Win; fp += 256) { g_last(fp, r0, nrows); g_sig(); g_term(); g_store(); }
            break;
        }
        bs += SPX; if (bs >= RPa) bs -= RPa;
        hb += 2 * TH; if (hb >= RING) hb -= RING;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) sred[w] = part;
    __syncthreads();
    if (tid == 0) a.val[mg] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

static int dec_bg_rpa(int Win) { return (128 + Win + 15) & ~15; }      // ring slots: the TH + 1 rows of a strip and the 128-pixel fetch of the next one
static size_t dec_bg_lds(int Win, int TH, int C) {
    const int RPa = dec_bg_rpa(Win), RING = 2 * TH + 2, NG = 3 * C;
    return (size_t)(RPa + 16) * 17 * sizeof(float4) + (size_t)RING * NG * (2 * Win + 16) * sizeof(float) + 128 * sizeof(float);
}
constexpr size_t DEC_BG_MAX_LDS = 96 * 1024;
int init_generic_dec_kernels() {
    if (hipFuncSetAttribute((const void*)k_convt_12<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONVT_12_MAX_LDS) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_dec_bg<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DEC_BG_MAX_LDS) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_dec_bg<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DEC_BG_MAX_LDS) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_dec_bg<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DEC_BG_MAX_LDS) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_convt_p<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONVT_P_MAX_LDS) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_convt_p<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONVT_P_MAX_LDS) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_convt_p<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONVT_P_MAX_LDS) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_convt_p<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONVT_P_MAX_LDS) != hipSuccess) return 1;
    return 0;
}
// the fused kernel's limits: C <= 3 channels, a full-width strip of 128 / Win rows in LDS (the caller runs the layers separately otherwise)
bool dec_bg_ok(int Hin, int Win, int C) {
    if (Win < 4 || Win > 64 || Hin < 2 || C < 1 || C > 3) return false;
    return dec_bg_lds(Win, 128 / Win, C) <= DEC_BG_MAX_LDS;
}
// 0 = launched; 1 = geometry outside the kernel's limits
int launch_dec_bg(DecBGArgs a, hipStream_t st) {
    if (!dec_bg_ok(a.Hin, a.Win, a.C)) return 1;
    a.TH = 128 / a.Win;
    a.RPa = dec_bg_rpa(a.Win);
    const size_t lds = dec_bg_lds(a.Win, a.TH, a.C);
    a.magicWin = (unsigned)((0x100000000ull + (unsigned)a.Win - 1) / (unsigned)a.Win);
    const dim3 grid((unsigned)a.rows), blk(256);
    if (a.C == 1) hipLaunchKernelGGL(k_dec_bg<1>, grid, blk, lds, st, a);
    else if (a.C == 2) hipLaunchKernelGGL(k_dec_bg<2>, grid, blk, lds, st, a);
    else hipLaunchKernelGGL(k_dec_bg<3>, grid, blk, lds, st, a);
    return 0;
}

}  // namespace efe
