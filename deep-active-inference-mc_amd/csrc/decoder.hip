// Fused decoder kernels for gfx950: the three ConvTranspose2d + ReLU layers, the final
// ConvTranspose2d(32,1) + Sigmoid and the per-image EFE reductions of
// /root/reference/src/torchmodel.py:120-127 (po_net.13..20), torchutils.py:26-37, torchmodel.py:210-212,289,292.
//
// One workgroup (4 waves) owns one decoder image; activations live in LDS between layers:
//
//   k_dec_a :  x4[16x16x64] --LDS--> ConvT(64,64,s1)+ReLU --LDS (in place)--> ConvT(64,64,s2)+ReLU --> y2[32x32x64] (HBM)
//   k_dec_b4:  y2 strips --LDS--> ConvT(64,32,s2)+ReLU (registers) --MFMA--> tap values of the 32->1 conv, horizontal sums in registers
//              --LDS ring of H planes--> vertical gather + sigmoid + entropy / reward reduction (+ optional image store)
//   k_dec_a_s / k_dec_b4<4>: the same kernels with an image over eight / four workgroups, for launches of <= 128 images
//
// LDS images are [pixel][17 float4 slots]: the 16 channel quads of a pixel plus one slot of padding, so that the ds_read_b128 of an
// MFMA B fragment (32 pixels x same quad) is bank-conflict free (16 consecutive pixels cover the 16 bank quads: 17 p mod 16 = p) and a
// fragment address is pixel base + an immediate (k_fc4's batch tile alone is still XOR-swizzled).  Weights are read as pre-packed A
// fragments straight from L2 (1 KiB coalesced per wave-load, shared by all workgroups).  fp32 MFMA (v_mfma_f32_32x32x2_f32 and the
// 4x4x1 16-block form): exact fp32 numerics.
#include "mfma_pipe.h"

namespace efe {

// ---------------------------------------------------------------------------------------------------------
// k_dec_a: ConvTranspose2d(64,64,3,s1,p1)+ReLU then ConvTranspose2d(64,64,3,s2,p1,op1)+ReLU, one image per WG.
// ---------------------------------------------------------------------------------------------------------
// The staged image takes DA_PS = 17 float4 slots per pixel (16 channel quads + 1 pad): 16 consecutive pixels cover the 16 bank quads
// (17 p mod 16 = p), and a fragment address is pixel base + constant -- the chunk offset is an immediate of the ds_read (the XOR swizzle
// of rounds 1-2 cost an XOR and an add per chunk and 11 % of the LDS cycles in bank conflicts).
constexpr int DA_PS = 17;
constexpr int DA_BIAS = 257 * DA_PS;       // float4 index of the two bias vectors behind the image + zero pixel
// Four waves of 64 features x 64 pixels each (NTW = 2 32-pixel tiles per wave; 256 VGPRs, 2 waves per SIMD).
__global__ void __launch_bounds__(256, 2) k_dec_a(const DecAArgs a) {
    constexpr int NTW = 2;
    constexpr int NTHR = 512 / NTW;
    constexpr int NPH = 2048 / NTHR;                  // float4s per thread of each image half
    extern __shared__ __attribute__((aligned(16))) float4 sm[];        // [257 pixels][DA_PS slots]; pixel 256 = zeros
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    const float4* W1 = reinterpret_cast<const float4*>(a.w1);
    const float4* W2 = reinterpret_cast<const float4*>(a.w2);

    if ((int)blockIdx.x >= a.rows) return;
    // Image schedule: the first two images of a workgroup are static (blockIdx, blockIdx + grid), the rest are claimed from a
    // ticket counter.  The two workgroups of a CU do not run at the same speed (the older one wins the MFMA arbitration,
    // ~300k vs ~365k cycles per image), so a static stride leaves the younger one alone -- at half the CU's throughput -- for
    // the last ~18 % of the kernel.  Tickets are fetched two images ahead by thread 0 and handed over through LDS.
    int* const slot = reinterpret_cast<int*>(sm + DA_BIAS + 32);
    if (tid == 0) slot[0] = 2 * (int)gridDim.x + atomicAdd(a.queue, 1);
    int nimg = (int)blockIdx.x + (int)gridDim.x;
    // next image, in flight during compute
    f32x4 pfa[NPH], pfb[NPH];
    f32x4* smv = reinterpret_cast<f32x4*>(sm);
    {
        const f32x4* X = reinterpret_cast<const f32x4*>(a.x4) + (size_t)blockIdx.x * 4096;
#pragma unroll
        for (int it = 0; it < NPH; ++it) pfa[it] = X[it * NTHR + tid];
#pragma unroll
        for (int it = 0; it < NPH; ++it) pfb[it] = X[(it + NPH) * NTHR + tid];
    }
    if (tid < 16) sm[256 * DA_PS + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    // biases live in LDS: a global bias load inside an epilogue forces s_waitcnt vmcnt(0), i.e. waits for every store
    // issued before it (vmcnt retires in order) and serialises the whole store stream
    if (tid < 16) sm[DA_BIAS + tid] = reinterpret_cast<const float4*>(a.b1)[tid];
    else if (tid < 32) sm[DA_BIAS + tid] = reinterpret_cast<const float4*>(a.b2)[tid - 16];

    for (int img = blockIdx.x; img < a.rows;) {
        // stage the 16x16x64 input image (64 KiB) into the swizzled LDS layout
        // per-image laundering of the thread index: stops hipcc hoisting ~40 loop-invariant staging / prefetch addresses out of
        // the image loop, which pushed the kernel over 256 VGPRs (spill reloads carry s_waitcnt vmcnt(0): they serialised
        // the prefetch loads and waited for every outstanding y2 store)
        int tl_ = tid; asm volatile("" : "+v"(tl_));
        // this wave's 64 pixels / input positions: image rows 4w .. 4w+3, two 32-pixel tiles of two rows each
        const int j = tl_ & 31, h = (tl_ >> 5) & 1;
        const int pcol = j & 15;
        const int prow0 = 2 * NTW * w + (j >> 4);                      // tile nt covers rows prow0 + 2*nt
        // a dead row of the call (efe_rows.mask) keeps the schedule -- barriers, ticket, the next image's prefetch -- and skips the work
        const bool live = row_live(a.live, img);
        if (live) {   // pixel it * (NTHR / 16) + (tid >> 4), quad tid & 15: one address register, immediate offsets
            const int sbase = (tl_ >> 4) * DA_PS + (tl_ & 15);
#pragma unroll
            for (int it = 0; it < NPH; ++it) smv[sbase + it * (NTHR / 16) * DA_PS] = pfa[it];
#pragma unroll
            for (int it = 0; it < NPH; ++it) smv[sbase + (it + NPH) * (NTHR / 16) * DA_PS] = pfb[it];
        }
        __syncthreads();
        const int nnimg = slot[0];                     // the image after nimg (written one iteration ago)
        int ticket = 0;
        if (tid == 0) ticket = 2 * (int)gridDim.x + atomicAdd(a.queue, 1);     // lands during the layer-1 contraction
        const bool more = nimg < a.rows;
        {   // request the next image now (clamped on the last pass: unconditional loads keep pf[] in registers)
            const f32x4* X = reinterpret_cast<const f32x4*>(a.x4) + (size_t)(more ? nimg : img) * 4096;
#pragma unroll
            for (int it = 0; it < NPH; ++it) pfa[it] = (X + it * NTHR)[tl_];
#pragma unroll
            for (int it = 0; it < NPH; ++it) pfb[it] = (X + (it + NPH) * NTHR)[tl_];
        }

        f32x16 acc[2][NTW];
        // the accumulators start at the bias (register e of tile mt holds channel 32mt + (e&3) + 8(e>>2) + 4h) and ReLU is one
        // integer max in the epilogue: each VALU instruction there costs ~19 cycles of wave time beside the other wave's MFMAs
        auto acc_init = [&](int boff) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 bb = sm[DA_BIAS + boff + mt * 8 + 2 * g4 + h];
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) { acc[mt][nt][4 * g4] = bb.x; acc[mt][nt][4 * g4 + 1] = bb.y; acc[mt][nt][4 * g4 + 2] = bb.z; acc[mt][nt][4 * g4 + 3] = bb.w; }
                }
        };
        // ---------------- layer 1: out[oh,ow] = sum_{kh,kw} in[oh+1-kh, ow+1-kw] . W[:, :, kh, kw] --------------
        if (live) {
        acc_init(0);
        tap_loop_pd<2, NTW, 1>(acc, 9, W1, sm, h, [&](int t, int (&bs)[NTW], int (&sw)[NTW], int& wt) {
            const int kh = t / 3, kw = t - kh * 3;
            wt = t;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int sy = prow0 + 2 * nt + 1 - kh, sx = pcol + 1 - kw;
                const bool ok = sy >= 0 && sy < 16 && sx >= 0 && sx < 16;
                const int sp = ok ? sy * 16 + sx : 256;
                bs[nt] = sp * DA_PS; sw[nt] = 0;
            }
        }, ConvWIdx{});
        }
        __syncthreads();                // every wave is done reading the input image (and slot[0])
        if (tid == 0) slot[0] = ticket;
        // bias + ReLU, written back IN PLACE as the input image of layer 2 (same padded layout)
        if (live)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int pix = 32 * NTW * w + 32 * nt + j;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c4 = mt * 8 + 2 * g4 + h;
                    float4 v;
                    v.x = relu_bits(acc[mt][nt][4 * g4 + 0]); v.y = relu_bits(acc[mt][nt][4 * g4 + 1]);
                    v.z = relu_bits(acc[mt][nt][4 * g4 + 2]); v.w = relu_bits(acc[mt][nt][4 * g4 + 3]);
                    sm[pix * DA_PS + c4] = v;
                }
        }
        __syncthreads();

        // ---------------- layer 2 (stride 2): 4 output parities, oh = 2*ih - 1 + kh ----------------------------
        float* Y = a.y2 + (size_t)img * (32 * 32 * 64);
#pragma unroll 1
        for (int par = 0; par < (live ? 4 : 0); ++par) {
            const int ph = par >> 1, pw = par & 1;
            acc_init(16);
            tap_loop_pd<2, NTW, 1>(acc, (1 + ph) * (1 + pw), W2, sm, h, ConvT2Addr<NTW, DA_PS>{ph, pw, prow0, 2, pcol, 16, 16, 256}, ConvWIdx{});
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                // y2 layout (float4 units): [parity][8 channel groups][256 input positions][2 quads] -- for one (mt, g4) the 64
                // lanes (position j of this wave's tile, quad h) write 1 KiB contiguous; the NHWC order would split every
                // store into 32 scattered 32-byte pieces
                float4* yp = reinterpret_cast<float4*>(Y) + (size_t)par * 4096 + ((2 * NTW * w + 2 * nt) * 16 + j) * 2 + h;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        float4 v;
                        v.x = relu_bits(acc[mt][nt][4 * g4 + 0]); v.y = relu_bits(acc[mt][nt][4 * g4 + 1]);
                        v.z = relu_bits(acc[mt][nt][4 * g4 + 2]); v.w = relu_bits(acc[mt][nt][4 * g4 + 3]);
                        yp[(mt * 4 + g4) * 512] = v;
                    }
            }
        }
        __syncthreads();                // every wave is done reading layer-1's image before the next one overwrites it
        img = nimg; nimg = nnimg;
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_dec_a_s: the same two layers for SMALL launches (<= 128 images: the one-episode planner), one image over EIGHT workgroups.
// Workgroup (image, p) owns the layer-2 input rows 2p, 2p + 1 (output rows 4p .. 4p + 3): it computes layer 1 for rows 2p .. 2p + 3
// (the stride-2 layer reads one row below its own) from the input rows 2p - 1 .. 2p + 4, keeps them in LDS and contracts layer 2
// for its row pair.  25 % more layer-1 work per image, an eighth of the latency (74 -> ~17 us per launch).  Every output element
// sees the operations of k_dec_a in the same order (bias as start value, taps 0..8, channel blocks 0..7): bit-identical results.
// Waves: layer 1: (feature tile mt = w >> 1, row pair nt = w & 1); layer 2: (mt = w >> 1, parities {(0,0), (1,1)} or {(0,1), (1,0)}).
// ---------------------------------------------------------------------------------------------------------
constexpr int DAS_IN = 6 * 16;                      // staged input pixels (+ a zero pixel)
constexpr int DAS_L1 = 4 * 16;                      // layer-1 pixels kept (+ a zero pixel)
constexpr int DAS_BIAS = (DAS_IN + 1 + DAS_L1 + 1) * DA_PS;
constexpr size_t DAS_LDS_BYTES = (DAS_BIAS + 32) * sizeof(float4);
__global__ void __launch_bounds__(256, 2) k_dec_a_s(const DecAArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 sm[];
    float4* sin = sm;                                // [6 rows][16][DA_PS], pixel 96 = zeros
    float4* sl1 = sm + (DAS_IN + 1) * DA_PS;         // [4 rows][16][DA_PS], pixel 64 = zeros
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int img = blockIdx.x >> 3, p = blockIdx.x & 7;
    if (!row_live(a.live, img)) return;
    const float4* W1 = reinterpret_cast<const float4*>(a.w1);
    const float4* W2 = reinterpret_cast<const float4*>(a.w2);
    {   // stage input rows 2p - 1 .. 2p + 4 (rows outside the image are zeros): 1536 float4, six per thread
        const float4* X = reinterpret_cast<const float4*>(a.x4) + (size_t)img * 4096;
        const int c4 = tid & 15, px0 = tid >> 4;
        float4 v[6];
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int pix = px0 + 16 * it, lr = pix >> 4, y = 2 * p - 1 + lr;
            v[it] = (y >= 0 && y < 16) ? X[(y * 16 + (pix & 15)) * 16 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < 6; ++it) sin[(px0 + 16 * it) * DA_PS + c4] = v[it];
    }
    if (tid < 16) { sin[DAS_IN * DA_PS + tid] = make_float4(0.f, 0.f, 0.f, 0.f); sl1[DAS_L1 * DA_PS + tid] = make_float4(0.f, 0.f, 0.f, 0.f); }
    if (tid < 16) sm[DAS_BIAS + tid] = reinterpret_cast<const float4*>(a.b1)[tid];
    else if (tid < 32) sm[DAS_BIAS + tid] = reinterpret_cast<const float4*>(a.b2)[tid - 16];
    __syncthreads();
    const int mt = w >> 1;
    f32x16 acc[1][1];
    auto acc_init = [&](int boff) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 bb = sm[DAS_BIAS + boff + mt * 8 + 2 * g4 + h];
            acc[0][0][4 * g4] = bb.x; acc[0][0][4 * g4 + 1] = bb.y; acc[0][0][4 * g4 + 2] = bb.z; acc[0][0][4 * g4 + 3] = bb.w;
        }
    };
    {   // ---- layer 1 for image rows 2p + 2 nt + (j >> 4), nt = w & 1
        const int nt = w & 1;
        const int orow = 2 * p + 2 * nt + (j >> 4), ocol = j & 15;
        acc_init(0);
        tap_loop_pd<1, 1, 2>(acc, 9, W1, sin, h, [&](int t, int (&bs)[1], int (&sw)[1], int& wt) {
            const int kh = t / 3, kw = t - kh * 3;
            wt = t;
            const int sy = orow + 1 - kh, sx = ocol + 1 - kw;
            const bool ok = sy >= 0 && sy < 16 && sx >= 0 && sx < 16;
            bs[0] = (ok ? (sy - (2 * p - 1)) * 16 + sx : DAS_IN) * DA_PS; sw[0] = 0;
        }, PackedWIdx{2, 8, mt});
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float4 v;
            v.x = relu_bits(acc[0][0][4 * g4 + 0]); v.y = relu_bits(acc[0][0][4 * g4 + 1]);
            v.z = relu_bits(acc[0][0][4 * g4 + 2]); v.w = relu_bits(acc[0][0][4 * g4 + 3]);
            sl1[(32 * nt + j) * DA_PS + mt * 8 + 2 * g4 + h] = v;
        }
    }
    __syncthreads();
    // ---- layer 2 (stride 2) for the input rows 2p, 2p + 1: two output parities per wave
    float* Y = a.y2 + (size_t)img * (32 * 32 * 64);
    const int vrows = min(4, 16 - 2 * p);            // layer-1 rows of the patch that exist in the image
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
        const int par = (w & 1) ? (k ? 2 : 1) : (k ? 3 : 0);
        const int ph = par >> 1, pw = par & 1;
        acc_init(16);
        tap_loop_pd<1, 1, 2>(acc, (1 + ph) * (1 + pw), W2, sl1, h, ConvT2Addr<1, DA_PS>{ph, pw, j >> 4, 0, j & 15, vrows, 16, DAS_L1}, PackedWIdx{2, 8, mt});
        float4* yp = reinterpret_cast<float4*>(Y) + (size_t)par * 4096 + ((2 * p) * 16 + j) * 2 + h;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float4 v;
            v.x = relu_bits(acc[0][0][4 * g4 + 0]); v.y = relu_bits(acc[0][0][4 * g4 + 1]);
            v.z = relu_bits(acc[0][0][4 * g4 + 2]); v.w = relu_bits(acc[0][0][4 * g4 + 3]);
            yp[(mt * 4 + g4) * 512] = v;
        }
    }
}

constexpr size_t DA_LDS_BYTES = (257 * DA_PS + 33) * sizeof(float4);
int init_dec_b_kernels();
// kernels that need more than the default 64 KiB of dynamic LDS: set once per device (called from efe_create)
int init_decoder_kernels() {
    if (hipFuncSetAttribute((const void*)k_dec_a, hipFuncAttributeMaxDynamicSharedMemorySize, DA_LDS_BYTES) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_dec_a_s, hipFuncAttributeMaxDynamicSharedMemorySize, DAS_LDS_BYTES) != hipSuccess) return 1;
    return init_dec_b_kernels();
}

void launch_dec_a(const DecAArgs& a, hipStream_t st) {
    if (a.parts == 8) {                                   // small launch: an image over eight workgroups
        hipLaunchKernelGGL(k_dec_a_s, dim3(a.rows * 8), dim3(256), DAS_LDS_BYTES, st, a);
        return;
    }
    const size_t lds = DA_LDS_BYTES;
    const int grid = a.rows < 512 ? a.rows : 512;         // persistent: 2 workgroups per CU
    hipLaunchKernelGGL(k_dec_a, dim3(grid), dim3(256), lds, st, a);       // 0.87 of the fp32 MFMA peak alone (19200 images)
}

// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// k_dec_b4: the same layer pair, INPUT-STATIONARY.  One wave owns one input row of the strip (32 positions) and ALL FOUR output
// parities of it.  The nine (kh, kw) taps of the stride-2 transposed conv read only four shifted views of the input,
//     shift (0,0): taps of parities (0,0) (0,1) (1,0) (1,1)      shift (0,+1): parities (0,1) (1,1)
//     shift (+1,0): parities (1,0) (1,1)                          shift (+1,+1): parity (1,1)
// so a B fragment (LDS) is read once per shift and chunk and feeds up to four INDEPENDENT accumulator chains (4 / 2 / 2 / 1
// weight fragments): 9 tap-tiles per wave for every wave (the parity-pair split of k_dec_b gives 5 and 4), 4 LDS reads per chunk
// instead of 9, and back-to-back MFMAs never wait on their own accumulator.  SR = 4 input rows per strip, 4 waves.
//
// The 32 -> 1 conv: tap planes by v_mfma_f32_16x16x1_4b on the accumulators (as in k_dec_b), with the taps placed in A rows
// {0-2, 4-6, 8-10} so that the 16-lane group kh of a wave holds the three kw taps of every pixel.  The horizontal part of the 3 x 3
// sum is then formed IN REGISTERS (the two column parities of a pixel are the same lane of two accumulators, its left / right
// neighbours one DPP row shift away):   H[kh][r][ow] = sum_kw T[kh,kw][r][ow + 1 - kw],
// and only the three H planes of a source row go through LDS (two ds_write_b64 per parity row instead of twenty ds_write_b32; the
// gather reads 3 values per pixel instead of 9).  The plane ring shrinks from 23.7 to 7.7 KiB: 49 KiB of LDS per workgroup.
// ---------------------------------------------------------------------------------------------------------
// Measured alternatives of this kernel (profiles/r2_decb4_variants.txt): persistent over images with a ticket queue (prologue 28k ->
// 0 cycles per image) and / or three workgroups per CU (166 VGPRs without spills after laundering the lane index per strip, tap
// weights re-read from LDS, channel halves merged with a cross-lane add: 50 KiB LDS) all land on the same 5.96-5.98 ms per 19200
// images; the per-strip index laundering alone costs 2-3 % at two waves.  The shader clock is at 2.38 GHz in steady state (it
// ramps from 2.05 GHz over the first four launches after idle): the kernel is not clock- or power-limited.
// Deferred gather: the gather of a strip (sigmoid + entropy / reward terms of its 2 SR output rows: ~130 VALU instructions per wave) is not done
// between the strip's two barriers -- where it is pure non-MFMA wave time -- but DURING THE NEXT STRIP'S CONTRACTION, in pieces
// placed between the MFMA groups of the unrolled channel-block steps (a wave's VALU instructions issue in the shadow of its own
// 64-cycle MFMAs).  The H-plane ring holds 4 SR + 2 rows (27.6 KiB) so that the next strip's planes do not overwrite rows still
// being gathered; the next strip's input is requested behind the contraction's last weight-fragment request (the tap phase covers
// its latency instead of the gather); the images of the deferred rows are stored after the strip barrier.  Per-thread summation
// order is the same as when every strip is gathered between its own barriers (0.812 of the fp32 MFMA peak in that form, 0.843 in this one).
template <int PARTS>        // 1 = one workgroup per image, 4 = four (small launches)
__global__ void __launch_bounds__(256, 2) k_dec_b4(const DecBArgs a) {
    constexpr int SR = 4, NW = 4, NTHR = 256;
    constexpr int DB_ZERO = (SR + 1) * 32;
    // float4 slots per input pixel: 16 channel quads + 1 pad.  Padded, not XOR-swizzled: 16 consecutive pixels still cover the 16 bank
    // quads (17 j mod 16 = j), and a fragment address is (pixel base + h) + a constant chunk offset -- an immediate of the ds_read, where
    // the swizzle cost the contraction an XOR / OR and an add per chunk (VALU work beside the other wave's MFMA stream)
    constexpr int DB_PS = 17;
    constexpr int DB_IN_F4 = (DB_ZERO + 1) * DB_PS;
    constexpr int DB_YROWS = 4 * SR + 2;
    constexpr int NPF = (SR + 1) * 512 / NTHR;        // 10 float4s of the input strip per thread
    constexpr int NS = 32 / SR;
    extern __shared__ __attribute__((aligned(16))) float4 sm[];
    float* sH = reinterpret_cast<float*>(sm + DB_IN_F4);       // [ring row][kh][64 output columns]
    __shared__ float4 sb3[8];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    // Small launches (a.parts == 4: <= 128 images, the one-episode planner's expansions and simulations) split an image over four
    // workgroups: quarter k owns the output rows gathered from strips 2k and 2k + 1 (their H planes need the last y3 row pair of strip
    // 2k - 1, so a quarter contracts that strip again as a halo) -- 3 of the 8 strips of latency instead of 8.  The per-image sum is
    // DEFINED quarter-wise, ((Q0 + Q1) + (Q2 + Q3)) with Q_k reduced over the workgroup on its own, so that one workgroup walking all
    // eight strips (a.parts == 1) and four workgroups walking three each produce the same bits: results stay a function of the noise
    // keys alone, whatever the launch size.
    constexpr int parts = PARTS;
    const int img = parts == 1 ? (int)blockIdx.x : (int)(blockIdx.x >> 2);
    const int qtr = parts == 1 ? 0 : (int)(blockIdx.x & 3);
    if (!row_live(a.live, img)) return;                // a dead row of the call (efe_rows.mask): workgroup-uniform
    const int s_lo = parts == 1 ? 0 : (qtr ? 2 * qtr - 1 : 0);          // first strip contracted (the halo strip of quarters 1..3)
    const int s_hi = parts == 1 ? NS : 2 * qtr + 2;

    const int mg = a.m0 + img;
    const int g = mg / a.rows_per_group;
    const int r = mg - g * a.rows_per_group;
    int gt, gp, gs;
    group_decode(a.gm, g, gt, gp, gs);
    const int mode = (gp == 0 && a.reward0) ? 1 : 0;
    const int slot = (gp == 0 && a.store0) ? gt * a.gm.S + gs : -1;
    float* po = (slot >= 0) ? a.po + ((size_t)slot * a.rows_per_group + r) * 4096 : nullptr;

    __shared__ float sq[4 * NTHR];       // [quarter][thread]: the threads' partial sums of a quarter, reduced once per image
    __shared__ float sQ[4];
    if (tid < 8) sb3[tid] = reinterpret_cast<const float4*>(a.b3)[tid];
    // 16-block 4x4x1 form of the tap contraction: block = 4 consecutive lanes = 4 pixels of one channel half, A row i = lane & 3 = kw,
    // one instruction per (kh, accumulator register): 3 x 16 A values per lane, 12 tap rows (9 used) instead of 16.  The 48 values depend
    // on (lane & 3, h) only: they live in an LDS table of 8 patterns and are re-read (12 broadcast ds_read_b128) in front of every
    // strip's tap phase instead of occupying 48 VGPRs through the contraction.
    __shared__ float4 sW4[8 * 12];
    if (tid < 96) {
        const int pat = tid / 12, i4 = tid - pat * 12;           // pattern = h * 4 + kw, float4 i4 = (kh, g4)
        const int kw = pat & 3, hh = pat >> 2, kh = i4 >> 2, g4 = i4 & 3;
        sW4[tid] = kw < 3 ? reinterpret_cast<const float4*>(a.w4 + (3 * kh + kw) * 32)[2 * g4 + hh] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4* w4p = sW4 + (h * 4 + (lane & 3)) * 12;
    if (tid < 16) sm[DB_ZERO * DB_PS + tid] = make_float4(0.f, 0.f, 0.f, 0.f);

    const float4* X = reinterpret_cast<const float4*>(a.y2) + (size_t)img * (32 * 32 * 16);
    const float4* W3 = reinterpret_cast<const float4*>(a.w3);
    const __amdgpu_buffer_rsrc_t wr = wrsrc(W3);
    const unsigned ln = (unsigned)lane * 16u;
    const float D1 = 1.00001f, D0 = 0.00001f;
    float part = 0.f;
    f32x4 pf[NPF];
    f32x4* smv = reinterpret_cast<f32x4*>(sm);
    const f32x4* Xv = reinterpret_cast<const f32x4*>(X);
    auto y2_at = [&](int iy, int idx) -> size_t { return (size_t)(((iy & 1) * 16 + ((idx & 511) >> 5)) * 512 + (iy >> 1) * 32 + (idx & 31)); };
#pragma unroll
    for (int it = 0; it < NPF; ++it) { const int idx = it * NTHR + tid; pf[it] = Xv[y2_at(min(SR * s_lo + (idx >> 9), 31), idx)]; }

    // the four shifted B views of this wave's row: LDS float4 base and swizzle key
    const int spA = w * 32 + j;                                   // (row w,     col j)
    const int spB = (j < 31) ? w * 32 + j + 1 : DB_ZERO;          // (row w,     col j + 1)   (column 32 does not exist)
    const int spC = (w + 1) * 32 + j;                             // (row w + 1, col j)
    const int spD = (j < 31) ? (w + 1) * 32 + j + 1 : DB_ZERO;    // (row w + 1, col j + 1)
    // packed-weight tap indices kh * 3 + kw: parity p = 2 * ph + pw
    //   shift A: p0 (1,1)=4  p1 (1,2)=5  p2 (2,1)=7  p3 (2,2)=8 | shift B: p1 (1,0)=3  p3 (2,0)=6 | shift C: p2 (0,1)=1  p3 (0,2)=2 | shift D: p3 (0,0)=0
    auto wf = [&](int tap, int kc) -> float4 { return wfrag(wr, ln, (size_t)(tap * 8 + kc) * 64); };

    // ---- gather of output row oh (lane = column): out = b4 + H[0][oh + 1] + H[1][oh] + H[2][oh - 1] (source row r contributes to
    // oh = r - 1 + kh), the two channel halves added here; split into pieces for the deferred form
    constexpr int RWG = 2 * SR / NW;
    float gh[RWG][6], gpr[RWG];
    // (ring slot of y3 row tr = slot hb0 of row r0y + (tr - r0y), wrapped: no division; a row outside the image reads any valid slot)
    auto g_load = [&](int oh, int q, int r0y, int hb0) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int tr = oh + 1 - kh;
            const bool rv = tr >= 0 && tr <= 63;
            int hs = hb0 + (tr - r0y);
            hs = hs < 0 ? hs + DB_YROWS : hs;
            hs = hs >= DB_YROWS ? hs - DB_YROWS : hs;
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
        // hardware exp2 / rcp / log2 (v_exp_f32, v_rcp_f32, v_log_f32: 1 ulp each) instead of the libm expansions: ~14 instead of ~60 VALU
        // instructions per pixel, each of which costs ~19 cycles of wave time beside the other wave's MFMA stream; the image stays within
        // 3e-7 of the libm form, the 4096-pixel sums within their own fp32 rounding (tests/test_gpu_parity.py tolerances unchanged)
        gpr[q] = hw_sigmoid(v);
    };
    auto g_term = [&](int oh, int q) {          // branch-free: both forms are evaluated (the reward form is four FMAs), rows above the image add zero
        // (the products are contracted EXPLICITLY: this lambda is inlined at several places -- deferred pieces, strip epilogues, both
        // template instances -- and a row must get the same bits whichever copy evaluates it; left to the compiler, a b - c d may
        // become fma(a, b, -(c d)) in one copy and fma(-c, d, a b) in another)
#pragma clang fp contract(off)
        const float pr = gpr[q];
        const float l1 = hw_log(D1 - pr), l0 = hw_log(D0 + pr);
        const float te = __builtin_fmaf(pr - 1.0f, l1, -(pr * l0));          // -(1 - p) ln((1e-5 + 1) - p) - p ln(1e-5 + p)
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
    auto prefetch = [&](int sn, int tl) {
#pragma unroll
        for (int it = 0; it < NPF; ++it) {
            const int idx = it * NTHR + tl;
            const int grow = min(SR * sn + (idx >> 9), 31);
            pf[it] = Xv[y2_at(grow, idx)];
        }
    };

    // a quarter is complete: park the thread's partial sum (one LDS write; the reductions of all quarters run once, behind the last strip)
    auto fold = [&](int k) { sq[k * NTHR + tid] = part; part = 0.f; };
    int hb = 0, hbp = 0;                                   // H-ring slots of y3 rows 2 SR s (this strip's first) and 2 SR (s - 1)
    for (int s = s_lo; s < s_hi; ++s) {
        const int tl = tid;
        const bool gq = parts == 1 ? true : s == 2 * qtr + 1;   // (uniform) the previous strip's rows are this workgroup's to gather
        const int oh0 = 2 * SR * (s - 1) - 1 + w, oh1 = oh0 + NW;         // this wave's two output rows of the previous strip        // (laundering the index per strip frees ~20 VGPRs -- 168, three waves per SIMD, 4 spills -- for no gain: 0.814 vs 0.812)
#pragma unroll
        for (int it = 0; it < NPF; ++it) {
            const int idx = it * NTHR + tl;
            const int rl = idx >> 9, seg = (idx & 511) >> 5, wi = idx & 31;
            const int ix = 2 * (wi >> 1) + (seg >> 3), c4 = 2 * (seg & 7) + (wi & 1);
            smv[(rl * 32 + ix) * DB_PS + c4] = (SR * s + rl < 32) ? pf[it] : (f32x4)(0.f);
        }
        float4 a0 = wf(4, 0), a1 = wf(5, 0), a2 = wf(7, 0), a3 = wf(8, 0);      // the strip's first weight fragments: in flight across the barrier
        __syncthreads();

        // the accumulators start at the layer-3 bias (register e holds channel (e & 3) + 8 (e >> 2) + 4 h): the bias vector is the C operand
        // of each chain's FIRST MFMA -- 16 short-lived registers instead of 64 moves into the four accumulator tiles
        f32x16 acc[4], bias16;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 bb = sb3[2 * g4 + h];
            bias16[4 * g4] = bb.x; bias16[4 * g4 + 1] = bb.y; bias16[4 * g4 + 2] = bb.z; bias16[4 * g4 + 3] = bb.w;
        }
        // ---- contraction, software-pipelined one chunk ahead: shift A (4 chains), shift B (2), shifts C + D fused (2 + 1, so the
        // single-chain shift never runs alone).  The first weight fragments were requested before the staging barrier.
        {
            float4 b = sm[spA * DB_PS + h];
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const float4 c0 = a0, c1 = a1, c2 = a2, c3 = a3, cb = b;
                if (kc < 7) {
                    a0 = wf(4, kc + 1); a1 = wf(5, kc + 1); a2 = wf(7, kc + 1); a3 = wf(8, kc + 1);
                    b = sm[spA * DB_PS + 2 * (kc + 1) + h];
                } else {
                    a0 = wf(3, 0); a1 = wf(6, 0);
                    b = sm[spB * DB_PS + h];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kc == 0) { MFMA4I(acc[0], bias16, c0, cb) MFMA4I(acc[1], bias16, c1, cb) MFMA4I(acc[2], bias16, c2, cb) MFMA4I(acc[3], bias16, c3, cb) }
                else { MFMA4(acc[0], c0, cb) MFMA4(acc[1], c1, cb) MFMA4(acc[2], c2, cb) MFMA4(acc[3], c3, cb) }
                // the previous strip's gather, a piece per channel-block step
                if (gq) {
                    if (kc == 0) { g_load(oh0, 0, 2 * SR * (s - 1), hbp); g_load(oh1, 1, 2 * SR * (s - 1), hbp); }
                    if (kc == 2) g_sig(oh0, 0);
                    if (kc == 4) g_term(oh0, 0);
                    if (kc == 6) g_sig(oh1, 1);
                }
            }
            float4 bd;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const float4 c0 = a0, c1 = a1, cb = b;
                if (kc < 7) {
                    a0 = wf(3, kc + 1); a1 = wf(6, kc + 1);
                    b = sm[spB * DB_PS + 2 * (kc + 1) + h];
                } else {
                    a0 = wf(1, 0); a1 = wf(2, 0); a2 = wf(0, 0);
                    b = sm[spC * DB_PS + h];
                    bd = sm[spD * DB_PS + h];
                }
                __builtin_amdgcn_sched_barrier(0);
                MFMA4(acc[1], c0, cb) MFMA4(acc[3], c1, cb)
                if (kc == 1 && gq) g_term(oh1, 1);
            }
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const float4 c0 = a0, c1 = a1, c2 = a2, cb = b, cd = bd;
                if (kc < 7) {
                    a0 = wf(1, kc + 1); a1 = wf(2, kc + 1); a2 = wf(0, kc + 1);
                    b = sm[spC * DB_PS + 2 * (kc + 1) + h];
                    bd = sm[spD * DB_PS + 2 * (kc + 1) + h];
                }
                if (kc == 6) prefetch((s < NS - 1) ? s + 1 : NS - 1, tl);       // the next strip's input, behind this strip's last weight-fragment request
                __builtin_amdgcn_sched_barrier(0);
                MFMA4(acc[3], c1, cb) MFMA4(acc[2], c0, cb) MFMA4(acc[3], c2, cd)
            }
        }
        // ---- ReLU, then the 32 -> 1 conv as tap planes: 16 x v_mfma_f32_16x16x1_4b per parity, the four parities' chains interleaved
        float w4g[3][16];
#pragma unroll
        for (int i4 = 0; i4 < 12; ++i4) {
            const float4 q = w4p[i4];
            w4g[i4 >> 2][4 * (i4 & 3)] = q.x; w4g[i4 >> 2][4 * (i4 & 3) + 1] = q.y; w4g[i4 >> 2][4 * (i4 & 3) + 2] = q.z; w4g[i4 >> 2][4 * (i4 & 3) + 3] = q.w;
        }
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            f32x4 Tq[2][3];                                  // [column parity][kh]: registers kw = 0..2 (3 = padding) of this lane's pixel and channel half
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[2 * ph][e] = relu_bits(acc[2 * ph][e]); acc[2 * ph + 1][e] = relu_bits(acc[2 * ph + 1][e]); }
#pragma unroll
            for (int pw = 0; pw < 2; ++pw)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) Tq[pw][kh] = (f32x4)(0.f);
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    Tq[0][kh] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4g[kh][e], acc[2 * ph][e], Tq[0][kh], 0, 0, 0);
                    Tq[1][kh] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4g[kh][e], acc[2 * ph + 1][e], Tq[1][kh], 0, 0, 0);
                }
            // horizontal presum per channel half (the halves are added by the gather): lane = (pixel c' = lane & 31, half h)
            int hsw = hb + 2 * w + ph;                       // ring slot of this wave's y3 row 2 (SR s + w) + ph
            hsw = hsw >= DB_YROWS ? hsw - DB_YROWS : hsw;
            float* hp = sH + ((hsw * 2 + h) * 3) * 64 + 2 * j;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                float l = wave_shr1(Tq[1][kh][2]), r = wave_shl1(Tq[0][kh][0]);
                l = (j == 0) ? 0.f : l;                       // lane 32 received lane 31 (the other channel half's pixel 31): image edge
                r = (j == 31) ? 0.f : r;
                float2 eo;
                eo.x = (Tq[1][kh][0] + Tq[0][kh][1]) + l;
                eo.y = (r + Tq[1][kh][1]) + Tq[0][kh][2];
                *reinterpret_cast<float2*>(hp + kh * 64) = eo;
            }
        }
        __syncthreads();
        if (gq) { g_store(oh0, 0); g_store(oh1, 1); }     // the deferred rows' pixels (stores behind the prefetch loads)
        if (parts == 1 && (s == 2 || s == 4 || s == 6)) fold(s / 2 - 1);       // quarter s / 2 - 1 is complete (its second gather ran inside this strip)
        if (s == s_hi - 1) {                                // no successor in this workgroup: the strip's own rows now (and row 63 behind the last strip)
#pragma unroll
            for (int q = 0; q <= RWG; ++q) {
                if (q == RWG && (w != 0 || s != NS - 1)) break;
                const int oh = 2 * SR * s - 1 + q * NW + w;
                g_load(oh, 0, 2 * SR * s, hb); g_sig(oh, 0); g_term(oh, 0); g_store(oh, 0);
            }
        }
        hbp = hb;
        hb += 2 * SR;
        hb = hb >= DB_YROWS ? hb - DB_YROWS : hb;
    }
    fold(parts == 1 ? 3 : 0);
    __syncthreads();
    // Q_k = the xor-tree sum over the 64 lanes of ((wave 0 + wave 1) + (wave 2 + wave 3)) of the parked partials: wave k reduces quarter k
    if (w < (parts == 1 ? 4 : 1)) {
        const float* qk = sq + w * NTHR + lane;
        float v = (qk[0] + qk[64]) + (qk[128] + qk[192]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) sQ[w] = v;
    }
    __syncthreads();
    if (tid == 0) {
        if (parts == 1) a.val[mg] = (sQ[0] + sQ[1]) + (sQ[2] + sQ[3]);
        else a.valq[(size_t)mg * 4 + qtr] = sQ[0];          // summed in the same association by the consumer (k_terms)
    }
}

constexpr size_t DB_LDS4D = ((5 * 32 + 1) * 17) * sizeof(float4) + 18 * 2 * 3 * 64 * sizeof(float);  // input strip + H planes per channel half (4 SR + 2 rows)
int init_dec_b_kernels() {
    if (hipFuncSetAttribute((const void*)k_dec_b4<1>, hipFuncAttributeMaxDynamicSharedMemorySize, DB_LDS4D) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_dec_b4<4>, hipFuncAttributeMaxDynamicSharedMemorySize, DB_LDS4D) != hipSuccess) return 1;
    return 0;
}

void launch_dec_b(const DecBArgs& a, hipStream_t st) {
    if (a.parts == 4) hipLaunchKernelGGL(k_dec_b4<4>, dim3(a.rows * 4), dim3(256), DB_LDS4D, st, a);
    else hipLaunchKernelGGL(k_dec_b4<1>, dim3(a.rows), dim3(256), DB_LDS4D, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// k_fc4: Linear(256, 64 * base^2) + ReLU + Dropout(0.5) (torchmodel.py:116-118; 16384 features for Dynamic dSprites, 28224 for the
// 84 x 84 geometry), output written NHWC (rows permuted at pack time).  A workgroup stages 64 batch rows x K=256 in swizzled LDS once
// and sweeps steps of 256 features; every wave owns 64 features x 64 rows per step.  Dropout mask = one Philox call per row per 128
// features.  The feature steps (ceil(mtiles / 8)) are dealt to 8 groups of SPG = ceil(steps / 8) consecutive steps (8 x 8 for 16384).
// ---------------------------------------------------------------------------------------------------------
__host__ __device__ inline int fc4_spg(int mtiles) { return ((mtiles + 7) / 8 + 7) / 8; }
template <int NT>          // 32-row batch tiles per workgroup tile: 2 (64 rows), or 1 for launches of <= 32 rows (the one-episode planner's trajectories)
__global__ void __launch_bounds__(256, 2) k_fc4(const GemmArgs a) {
    constexpr int RT = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) float4 sm[];        // [RT rows][64 quads], quad ^= row & 15
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    // Persistent, XCD-aware and balanced: workgroup b runs on XCD b % 8 and only ever touches feature group b & 7, so every XCD
    // streams ONE 2 MiB weight slice (L2-resident, 4 MiB per XCD).  The (row tile, feature step) pairs of a feature group are split
    // into equal contiguous ranges over the gridDim/8 workgroups of that XCD: with one workgroup per (row tile, group) the 2400
    // 230-us workgroups of a 19200-row launch ran as 4.7 waves over the 512 slots and the last, 70 %-full wave cost ~8 %.
    const int fgrp = blockIdx.x & 7;
    const int nper = gridDim.x >> 3, k = blockIdx.x >> 3;
    const int SPG = fc4_spg(a.mtiles);
    const int nsteps = ((a.n_pix + RT - 1) / RT) * SPG;                // (row tile, step) pairs of this feature group
    const int q0 = (int)(((long)nsteps * k) / nper), q1 = (int)(((long)nsteps * (k + 1)) / nper);
    f32x4* smv = reinterpret_cast<f32x4*>(sm);
    const float4* Wl = reinterpret_cast<const float4*>(a.Wp);
    auto xaddr = [&](int t, int (&bs)[NT], int (&sw)[NT], int& wt) {
        wt = t;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { bs[nt] = (nt * 32 + j) * 64 + t * 16; sw[nt] = j & 15; }
    };
    int cur_rt = -1;
    uint32_t krow[2] = {0, 0}, kstream[2] = {0, 0}, kstage[2] = {0, 0};
    bool rv[2] = {false, false};
#pragma unroll 1
    for (int q = q0; q < q1; ++q) {
        const int rt = q / SPG, fs = q - rt * SPG;
        const int row0 = rt * RT;
        if (rt != cur_rt) {                                            // (re)stage the 64-row tile: at most twice more than once per workgroup
            if (cur_rt >= 0) __syncthreads();                          // every wave is done reading the previous tile
            cur_rt = rt;
            const f32x4* X = reinterpret_cast<const f32x4*>(a.X);
#pragma unroll
            for (int it = 0; it < 8 * NT; ++it) {
                const int idx = it * 256 + tid;                        // RT rows x 64 quads
                const int r = idx >> 6, c4 = idx & 63;
                const int gr = row0 + r;
                smv[r * 64 + (c4 ^ (r & 15))] = (gr < a.n_pix) ? X[(size_t)gr * 64 + c4] : (f32x4)(0.f);
            }
            __syncthreads();
            // dropout keys of this lane's two rows
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int m = row0 + nt * 32 + j;
                rv[nt] = m < a.n_pix;
                const int mg = a.m0 + (rv[nt] ? m : 0);
                const int g = mg / a.rows_per_group;
                krow[nt] = global_row(a.gm.ids, a.gm.ids_div, mg - g * a.rows_per_group, a.row_offset);
                const uint2 key = group_key(a.gm, g);
                kstream[nt] = key.x; kstage[nt] = key.y;
            }
        }
        const int mt0 = (fgrp * SPG + fs) * 8 + 2 * w;                 // this wave's first 32-feature tile
        if (mt0 >= a.mtiles) continue;                                 // wave-uniform: past the last feature tile (mtiles is even)
        f32x16 acc[2][NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
        // the step's bias quads are requested BEFORE the contraction: a load placed between the stores would need
        // s_waitcnt vmcnt(0) and wait for every store ahead of it
        float4 bq[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) bq[mt][g4] = *reinterpret_cast<const float4*>(a.bias + (mt0 + mt) * 32 + 8 * g4 + 4 * h);
        tap_loop<2, NT>(acc, 4, Wl, sm, h, xaddr, DenseWIdx{mt0});
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (!rv[nt]) continue;
            const uint4 rnd = noise_words(a.k0, a.k1, a.tag, (uint32_t)((mt0 * 32) >> 7), krow[nt], kstream[nt], kstage[nt]);
            float* yp = a.Y + (size_t)(row0 + nt * 32 + j) * a.ldy;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int co = (mt0 + mt) * 32 + 8 * g4 + 4 * h;
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

void launch_fc4(const GemmArgs& a, hipStream_t st) {
    // persistent: 2 workgroups per CU (64 KiB LDS each), 8 feature groups x (up to) 64 workgroups, each with >= 1 (row tile, step) pair
    if (a.n_pix <= 32) {               // one 32-row tile: half the MFMA work of a 64-row tile whose second half would be padding
        const int nsteps = fc4_spg(a.mtiles);
        hipLaunchKernelGGL(k_fc4<1>, dim3(8 * (nsteps < 64 ? nsteps : 64)), dim3(256), 32 * 64 * sizeof(float4), st, a);
        return;
    }
    const int nsteps = ((a.n_pix + 63) / 64) * fc4_spg(a.mtiles);
    const int nper = nsteps < 64 ? nsteps : 64;
    hipLaunchKernelGGL(k_fc4<2>, dim3(8 * nper), dim3(256), 64 * 64 * sizeof(float4), st, a);
}

}  // namespace efe
