// Fused small-MLP kernels for gfx950: the transition net ModelMid.ps_net (/root/reference/src/torchmodel.py:41-52, 58-61) as ONE
// launch per stage, and the whole habit-policy rollout of mcts_step_simulate (torchmodel.py:360-390: encode_s -> sample an action
// -> transition -> reparameterise, `depth` times) as ONE launch.
//
// A workgroup (4 waves) owns 16 batch rows and walks them through every layer; activations never leave LDS.  The contraction
// uses v_mfma_f32_16x16x4_f32 (exact fp32, same FLOP rate as the 32x32x2 form): N = 16 batch rows, so a 2560-row stage is 160
// workgroups (a 32-row tile would leave 2/3 of the CUs idle), M = 16 features per tile, eight tiles (128 features = one Philox
// block of the MC-dropout mask) per wave.
//
//   weights : packed [16-feature tile][16-channel chunk][64 lanes][4]; lane l = (m = l & 15, q = l >> 4) holds
//             W[16 mt + m][16 kc + 4 q + s], s = 0..3  -- one coalesced 1 KiB buffer load per (tile, chunk), read from L2
//   B       : activations in LDS, [row n][512 floats], 16-byte quads XOR-swizzled by the row; lane (n = l & 15, q) reads the quad
//             16 kc + 4 q .. + 3 with one ds_read_b128; MFMA step s of a chunk contracts channels {16 kc + 4 q + s : q = 0..3}
//   D       : lane (n, q) holds features 16 mt + 4 q + 0..3 of row n: bias, ReLU, dropout and ONE ds_write_b128 back to LDS
//
// The summation order over K is fixed by the layer (never by the batch), so a row's result does not depend on which rows share
// its launch -- the property the multi-GPU / chunking invariance tests pin.
#include "kernels.h"

namespace efe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int FR = 16;               // batch rows per workgroup
constexpr int ACT_F4 = FR * 128;     // one activation buffer: 16 rows x 128 quads (512 floats)

__device__ __forceinline__ int aswz(int n, int c4) { return n * 128 + ((c4 & ~15) | ((c4 ^ n) & 15)); }

__device__ __forceinline__ float4 wfrag16(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned f4_index) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, f4_index * 16u, 0);
    return __builtin_bit_cast(float4, v);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc16(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}

#define MFMA16(ACC, AV, BV)                                                     \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).x, (BV).x, ACC, 0, 0, 0);   \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).y, (BV).y, ACC, 0, 0, 0);   \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).z, (BV).z, ACC, 0, 0, 0);   \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).w, (BV).w, ACC, 0, 0, 0);

// acc[mt] += W[tiles mt0 .. mt0 + NMT) x act over KC 16-channel chunks.  Weight fragments (L2, ~1 us under load) are requested two
// chunks ahead, the LDS activation fragment one chunk ahead, through THREE statically indexed register buffers (a two-buffer
// rotation makes hipcc copy every fragment, and the copy waits for the load it was meant to hide); consecutive MFMAs go to
// DIFFERENT accumulators (the 16x16x4 form has a 40-cycle dependent latency against a 32-cycle issue).
#define G16_STEP(AU, AL, BU, BL, K)                                                                                          \
    {                                                                                                                        \
        const int ka_ = (K) + 2 < KC ? (K) + 2 : KC - 1, kb_ = (K) + 1 < KC ? (K) + 1 : KC - 1;                              \
        _Pragma("unroll") for (int mt = 0; mt < NMT; ++mt) AL[mt] = wfrag16(wr, ln, (unsigned)((mt0 + mt) * KC + ka_) * 64u); \
        BL = act[aswz(n, 4 * kb_ + q)];                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        _Pragma("unroll") for (int mt = 0; mt < NMT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(AU[mt].x, BU.x, acc[mt], 0, 0, 0); \
        _Pragma("unroll") for (int mt = 0; mt < NMT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(AU[mt].y, BU.y, acc[mt], 0, 0, 0); \
        _Pragma("unroll") for (int mt = 0; mt < NMT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(AU[mt].z, BU.z, acc[mt], 0, 0, 0); \
        _Pragma("unroll") for (int mt = 0; mt < NMT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(AU[mt].w, BU.w, acc[mt], 0, 0, 0); \
    }
template <int NMT>
__device__ __forceinline__ void gemm16(f32x4 (&acc)[NMT], const float4* __restrict__ Wp, int mt0, int KC, const float4* act, int n, int q,
                                       unsigned ln) {
    const __amdgpu_buffer_rsrc_t wr = rsrc16(Wp);
    float4 a0[NMT], a1[NMT], a2[NMT], b0, b1, b2;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) a0[mt] = wfrag16(wr, ln, (unsigned)((mt0 + mt) * KC) * 64u);
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) a1[mt] = wfrag16(wr, ln, (unsigned)((mt0 + mt) * KC + (KC > 1 ? 1 : 0)) * 64u);
    b0 = act[aswz(n, q)];
    int kc = 0;
    for (; kc + 3 <= KC; kc += 3) {
        G16_STEP(a0, a2, b0, b1, kc)
        G16_STEP(a1, a0, b1, b2, kc + 1)
        G16_STEP(a2, a1, b2, b0, kc + 2)
    }
    if (kc < KC) G16_STEP(a0, a2, b0, b1, kc)              // KC % 3 tail (uniform branches)
    if (kc + 1 < KC) G16_STEP(a1, a0, b1, b2, kc + 1)
}
#undef G16_STEP

struct RowKey { uint32_t row, stream, stage; };

// hidden layer of one wave: features 16 * mt0 .. + 16 * NMT, bias (+ ReLU) (+ MC-dropout from one Philox block) -> LDS
template <int NMT, bool RELU, bool DROP>
__device__ __forceinline__ void hidden16(const float4* __restrict__ Wp, const float* __restrict__ bias, int mt0, int KC, const float4* act_in,
                                         float4* act_out, int n, int q, unsigned ln, uint32_t k0, uint32_t k1, uint32_t tag, const RowKey& rk) {
    f32x4 acc[NMT];
    float4 bq[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        acc[mt] = (f32x4)(0.f);
        bq[mt] = *reinterpret_cast<const float4*>(bias + 16 * (mt0 + mt) + 4 * q);
    }
    gemm16<NMT>(acc, Wp, mt0, KC, act_in, n, q, ln);
    uint4 rnd = make_uint4(0u, 0u, 0u, 0u);
    if (DROP) rnd = noise_words(k0, k1, tag, (uint32_t)((16 * mt0) >> 7), rk.row, rk.stream, rk.stage);     // NMT = 8: one 128-feature block per wave
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        const int f = 16 * (mt0 + mt) + 4 * q;
        float v[4] = {acc[mt][0] + bq[mt].x, acc[mt][1] + bq[mt].y, acc[mt][2] + bq[mt].z, acc[mt][3] + bq[mt].w};
        if (RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
        }
        if (DROP) {
            const int wsel = (f >> 5) & 3;
            const uint32_t word = wsel == 0 ? rnd.x : wsel == 1 ? rnd.y : wsel == 2 ? rnd.z : rnd.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((word >> ((f + e) & 31)) & 1u) ? v[e] * 2.0f : 0.0f;
        }
        act_out[aswz(n, 4 * (mt0 + mt) + q)] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// the four layers of ps_net for the workgroup's 16 rows: x (in bufA[n][0..15]) -> out [n][0..31] in bufA (features 0..19 real)
__device__ __forceinline__ void trans_chain(const MlpW& W, float4* bufA, float4* bufB, int w, int n, int q, unsigned ln, uint32_t k0,
                                            uint32_t k1, const RowKey& rk) {
    hidden16<8, true, true>(W.w[0], W.b[0], 8 * w, 1, bufA, bufB, n, q, ln, k0, k1, TAG_MID + 0, rk);
    __syncthreads();
    hidden16<8, true, true>(W.w[1], W.b[1], 8 * w, 32, bufB, bufA, n, q, ln, k0, k1, TAG_MID + 1, rk);
    __syncthreads();
    hidden16<8, true, true>(W.w[2], W.b[2], 8 * w, 32, bufA, bufB, n, q, ln, k0, k1, TAG_MID + 2, rk);
    __syncthreads();
    if (w < 2) hidden16<1, false, false>(W.w[3], W.b[3], w, 32, bufB, bufA, n, q, ln, k0, k1, 0u, rk);
    __syncthreads();
}

// ---- the same chain with the two 512 x 512 layers SPLIT OVER NWG WORKGROUPS (k_sim_chain<NWG>, one-episode decisions) -------------------
// Workgroup kw of a group computes feature tiles 4 kw .. 4 kw + 3 of layers 2 and 3 (one 16-feature tile per wave: an eighth of the weight
// stream that bounds the one-workgroup form) and everything else redundantly.  The slices are exchanged through a small global buffer with
// AGENT-SCOPE accesses (sc1: write-through stores, cache-bypassing loads) and a counter -- no release / acquire fences: on gfx950 those
// write back and invalidate the XCD's whole L2 (the fenced form of round 4 was slower than one workgroup).  Every feature is contracted by
// the same MFMA sequence over K as in trans_chain: bit-identical results.
constexpr unsigned AUX_SC1 = 16u;                 // cache-policy bit 4 of the raw buffer intrinsics = sc1 on gfx940+ (agent scope: write-through stores, loads that miss the
                                                  // XCD-local L2); the engine enables the split chain on gfx950 only (efe_create_cfg checks gcnArchName)
constexpr int XCH_ROW_F4 = 128;                   // exchange buffer: [16 rows][512 floats]

// one 16-feature tile over KC = 32 chunks with the weight fragments EIGHT chunks ahead (a ring of eight statically indexed registers, the
// loop fully unrolled): a single tile's four MFMAs per chunk (128 cycles) cannot cover the L2 round trip two chunks ahead, and in the split
// chain nothing else runs on the SIMD.  Same accumulation order over K as gemm16<1>: bit-identical.
__device__ __forceinline__ void gemm16_deep(f32x4& acc, const float4* __restrict__ Wp, int mt0, const float4* act, int n, int q, unsigned ln) {
    constexpr int KC = 32, PD = 8;
    const __amdgpu_buffer_rsrc_t wr = rsrc16(Wp);
    float4 af[PD], bf[2];
#pragma unroll
    for (int p = 0; p < PD; ++p) af[p] = wfrag16(wr, ln, (unsigned)(mt0 * KC + p) * 64u);
    bf[0] = act[aswz(n, q)];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        const float4 av = af[kc % PD], bv = bf[kc & 1];
        if (kc + PD < KC) af[kc % PD] = wfrag16(wr, ln, (unsigned)(mt0 * KC + kc + PD) * 64u);
        if (kc + 1 < KC) bf[(kc + 1) & 1] = act[aswz(n, 4 * (kc + 1) + q)];
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
    }
}

template <bool RELU, bool DROP>
__device__ __forceinline__ float4 hidden16_slice(const float4* __restrict__ Wp, const float* __restrict__ bias, int mt0, int KC, const float4* act_in,
                                                 int n, int q, unsigned ln, uint32_t k0, uint32_t k1, uint32_t tag, const RowKey& rk) {
    f32x4 acc[1];
    acc[0] = (f32x4)(0.f);
    const float4 bq = *reinterpret_cast<const float4*>(bias + 16 * mt0 + 4 * q);
    gemm16_deep(acc[0], Wp, mt0, act_in, n, q, ln);          // (KC = 32: the two 512-wide layers)
    const int f = 16 * mt0 + 4 * q;
    float v[4] = {acc[0][0] + bq.x, acc[0][1] + bq.y, acc[0][2] + bq.z, acc[0][3] + bq.w};
    if (RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
    }
    if (DROP) {
        const uint4 rnd = noise_words(k0, k1, tag, (uint32_t)(f >> 7), rk.row, rk.stream, rk.stage);
        const int wsel = (f >> 5) & 3;
        const uint32_t word = wsel == 0 ? rnd.x : wsel == 1 ? rnd.y : wsel == 2 ? rnd.z : rnd.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((word >> ((f + e) & 31)) & 1u) ? v[e] * 2.0f : 0.0f;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

// all NWG workgroups of the group have stored their slice of xbuf -> the full [16][512] activation in LDS (act_out).
// A peer that does not arrive within the spin bound (~0.3 s) must never produce a silently wrong rollout: the workgroup that timed out
// raises the group's STICKY flag (sync[2], agent scope) before it goes on -- i.e. before it publishes any further slice or arrival -- and
// the writer reads the flag behind its last gather (k_sim_chain's epilogue): whichever workgroup saw the timeout, at whichever exchange,
// the writer poisons every output of the launch.  The words re-arm themselves at the end of a launch; after a call that failed the
// engine zeroes them on the stream before the next split launch (engine.hip, sim_sync_dirty), so a launch that was cut short cannot
// leave the next one's barriers open.
template <int NWG>
__device__ __forceinline__ void xchg_gather(const float* xbuf, int* sync, int target, float4* act_out, int tid, int* bad) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's slice stores are acknowledged (write-through)
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 18)) __builtin_amdgcn_s_sleep(1);
        // (returning form: the flag is set before this workgroup's next arrival, and every arrival of the LAST exchange precedes the
        // writer's read of the flag in the epilogue -- one read per launch, not one per gather: an agent-scope round trip costs ~1 us)
        if (spins >= (1 << 18)) *bad = 1 | __hip_atomic_fetch_or(sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t xr = rsrc16(xbuf);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * 256 + tid, row = idx >> 7, c4 = idx & 127;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)idx * 16u, 0, AUX_SC1);
        act_out[aswz(row, c4)] = __builtin_bit_cast(float4, v);
    }
    __syncthreads();
}

template <int NWG>
__device__ __forceinline__ void trans_chain_x(const MlpW& W, float4* bufA, float4* bufB, int w, int n, int q, unsigned ln, uint32_t k0, uint32_t k1,
                                              const RowKey& rk, int kw, float* xch, int* sync, int& xn, int tid, int* bad) {
    hidden16<8, true, true>(W.w[0], W.b[0], 8 * w, 1, bufA, bufB, n, q, ln, k0, k1, TAG_MID + 0, rk);        // K = 16: every workgroup computes all of it
    __syncthreads();
    const int mt0 = 4 * kw + w;
    {
        const float4 v = hidden16_slice<true, true>(W.w[1], W.b[1], mt0, 32, bufB, n, q, ln, k0, k1, TAG_MID + 1, rk);
        float* xb = xch + (size_t)(xn & 1) * (16 * 512);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc16(xb), (unsigned)((n * XCH_ROW_F4 + 4 * mt0 + q) * 16), 0, AUX_SC1);
        xchg_gather<NWG>(xb, sync, NWG * (xn + 1), bufA, tid, bad);
        ++xn;
    }
    {
        const float4 v = hidden16_slice<true, true>(W.w[2], W.b[2], mt0, 32, bufA, n, q, ln, k0, k1, TAG_MID + 2, rk);
        float* xb = xch + (size_t)(xn & 1) * (16 * 512);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc16(xb), (unsigned)((n * XCH_ROW_F4 + 4 * mt0 + q) * 16), 0, AUX_SC1);
        xchg_gather<NWG>(xb, sync, NWG * (xn + 1), bufB, tid, bad);
        ++xn;
    }
    if (w < 2) bufA[aswz(n, 4 * w + q)] = hidden16_slice<false, false>(W.w[3], W.b[3], w, 32, bufB, n, q, ln, k0, k1, 0u, rk);      // 512 -> 20 (+ padding), deep prefetch
    __syncthreads();
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// k_trans_fused: ModelMid.ps_net over M rows ([group][row] batch; every group may read the same x rows, x_mod).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) k_trans_fused(const TransFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 sm[];
    float4* bufA = sm;
    float4* bufB = sm + ACT_F4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const unsigned ln = (unsigned)lane * 16u;
    const int row0 = blockIdx.x * FR;
    {   // x rows -> bufA[n][quads 0..3]
        if (tid < 64) {
            const int m = row0 + n;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < a.M) {
                const int r = a.x_mod > 0 ? m % a.x_mod : m;
                v = reinterpret_cast<const float4*>(a.X + (size_t)r * 16)[q];
            }
            bufA[aswz(n, q)] = v;
        }
    }
    RowKey rk;
    {
        const int m = min(row0 + n, a.M - 1);
        const int mg = a.m0 + m;
        const int g = mg / a.rows_per_group;
        rk.row = global_row(a.gm.ids, a.gm.ids_div, mg - g * a.rows_per_group, a.row_offset);
        const uint2 key = group_key(a.gm, g);
        rk.stream = key.x; rk.stage = key.y;
    }
    __syncthreads();
    trans_chain(a.W, bufA, bufB, w, n, q, ln, a.k0, a.k1, rk);
    if (tid < 64) {                       // tr[m][32]: mean 0..9, logvar 10..19 (20..31 are the zero-padded features)
        const int m = row0 + n;
        if (m < a.M) {
            reinterpret_cast<float4*>(a.tr + (size_t)m * 32)[q] = bufA[aswz(n, q)];
            reinterpret_cast<float4*>(a.tr + (size_t)m * 32)[4 + q] = bufA[aswz(n, 4 + q)];
        }
    }
}

void launch_trans_fused(const TransFusedArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_trans_fused, dim3((a.M + FR - 1) / FR), dim3(256), 2 * ACT_F4 * sizeof(float4), st, a);
}

// ---------------------------------------------------------------------------------------------------------
// k_sim_chain: the habit-policy rollout of mcts_step_simulate for 8 episodes per workgroup, all `T` steps in one launch:
//   q = softmax(qpi_net(s_t));  a_t ~ q (inverse CDF on a Philox / injected uniform; invalid q -> action 0, the reference's bare
//   except);  (mean, logvar) = ps_net([onehot(a_t) | s_t]) with MC-dropout;  ps1 = eps * exp(logvar / 2) + mean;
//   s_{t+1} = use_means ? mean : ps1.   Trajectory arrays are [E][T][...] (rows of the trajectory batch that follows).
// The 16-row MFMA tile carries every episode TWICE: rows 0..7 are the rollout's own transition (pass SIM, sample t, row = episode),
// rows 8..15 the loop-2 transition of calculate_G_given_trajectory on the same input (/root/reference/src/torchmodel.py:339-341:
// pass T2, row = episode * T + t) -- the same weight stream, different dropout keys.  Both land in `tr` in the layout of the
// trajectory core ([2 groups][E * T rows][32]: group 0 = the given (mean, logvar), group 1 = the T2 transition), which then needs no
// transition launch of its own (one-episode decisions: 39 us of a 0.6 ms iteration, on the critical path).
// ---------------------------------------------------------------------------------------------------------
// NWG = 1: one workgroup per 8 episodes (any batch size).  NWG = 8: the same group of 8 episodes on EIGHT workgroups that split the two wide
// layers of the transition net (trans_chain_x): 237 -> ~100 us per launch for the one-episode planner, whose critical path it is.
template <int NWG>
__global__ void __launch_bounds__(256, 1) k_sim_chain(const SimChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 sm[];
    float4* bufA = sm;
    float4* bufB = sm + ACT_F4;
    float* srow = reinterpret_cast<float*>(sm + 2 * ACT_F4);        // [8 episodes][16]: current state s_t (10 used)
    float* sact = srow + SIM_FE * 16;                               // [8 episodes][8]: one-hot action of the step
    const int A = a.pi_dim;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int ne = n & 7, second = n >> 3;                          // tile row n = (episode slot ne, pass: 0 = SIM, 1 = T2)
    const unsigned ln = (unsigned)lane * 16u;
    const int grp = NWG == 1 ? (int)blockIdx.x : (int)blockIdx.x / NWG, kw = NWG == 1 ? 0 : (int)blockIdx.x % NWG;
    const int e0 = grp * SIM_FE;
    const int E = a.E, T = a.T;
    const uint32_t stage_ = a.stage;
    const bool writer = kw == 0;                                    // (every workgroup of a group computes the same rollout; one writes it)
    __shared__ int bad_;
    if (tid == 0) bad_ = 0;
    int xn = 0;                                                     // exchanges done (NWG > 1)
    float* const xch = NWG == 1 ? nullptr : a.xch + (size_t)grp * (2 * 16 * 512);
    int* const sync = NWG == 1 ? nullptr : a.sync + grp * 4;
    const uint32_t erow = global_row(a.ids, 1, min(e0 + ne, E - 1), a.row_offset);      // (episodes past E: any valid key, results discarded)
    if (tid < SIM_FE * 16) {
        const int rr = tid >> 4, k = tid & 15, e = e0 + rr;
        srow[tid] = (k < 10 && e < E) ? a.s0[(size_t)e * 10 + k] : 0.f;
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        // ---- habit net (no dropout): x = [s | 0] -> 128 -> 128 -> 4
        if (tid < 64) bufA[aswz(n, q)] = reinterpret_cast<const float4*>(srow + ne * 16)[q];
        __syncthreads();
        RowKey none{0u, 0u, 0u};
        hidden16<2, true, false>(a.H.w[0], a.H.b[0], 2 * w, 1, bufA, bufB, n, q, ln, 0u, 0u, 0u, none);
        __syncthreads();
        hidden16<2, true, false>(a.H.w[1], a.H.b[1], 2 * w, 8, bufB, bufA, n, q, ln, 0u, 0u, 0u, none);
        __syncthreads();
        if (w == 0) hidden16<1, false, false>(a.H.w[2], a.H.b[2], 0, 8, bufA, bufB, n, q, ln, 0u, 0u, 0u, none);
        __syncthreads();
        // ---- softmax (torchmodel.py:28-29), categorical sample (torchmodel.py:364,379), one thread per episode
        if (tid < SIM_FE) {
            const int e = e0 + tid;
            const float4 lg = bufB[aswz(tid, 0)], lg2 = bufB[aswz(tid, 1)];
            const float l[8] = {lg.x, lg.y, lg.z, lg.w, lg2.x, lg2.y, lg2.z, lg2.w};
            float mx = -INFINITY;
            for (int k = 0; k < A; ++k) mx = fmaxf(mx, l[k]);
            float ex[8], sum = 0.f;
            for (int k = 0; k < A; ++k) { ex[k] = expf(l[k] - mx); sum += ex[k]; }
            float qq[8], tot = 0.f; bool bad = false;
            for (int k = 0; k < A; ++k) { qq[k] = ex[k] / sum; if (!(qq[k] >= 0.f) || isinf(qq[k])) bad = true; tot += qq[k]; }
            int act = 0;
            if (!bad && tot > 0.f) {
                const float u = a.u_inj ? a.u_inj[(size_t)t * E + min(e, E - 1)]
                                        : u01(noise_words(a.k0, a.k1, TAG_ACT, 0u, global_row(a.ids, 1, min(e, E - 1), a.row_offset), stream_id(PASS_HABIT, (uint32_t)t), stage_).x);
                const float thr = u * tot;
                float accq = 0.f; act = A - 1;
                for (int k = 0; k < A; ++k) { accq += qq[k]; if (thr < accq) { act = k; break; } }
            } else bad = true;
            for (int k = 0; k < A; ++k) {
                const float oh = (k == act) ? 1.f : 0.f;
                sact[tid * 8 + k] = oh;
                if (e < E && writer) {
                    a.pi0[((size_t)e * T + t) * A + k] = oh;
                    if (t == 0 && a.Qpi0) a.Qpi0[(size_t)e * A + k] = bad ? oh : qq[k];
                }
            }
        }
        __syncthreads();
        // ---- transition: x = [pi | s | 0 0] (torchmodel.py:59), the same input on the episode's two tile rows
        if (tid < 64) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 4 * q + i;
                v[i] = k < A ? sact[ne * 8 + k] : k < A + 10 ? srow[ne * 16 + (k - A)] : 0.f;
            }
            bufA[aswz(n, q)] = make_float4(v[0], v[1], v[2], v[3]);
        }
        const RowKey rk = second ? RowKey{erow * (uint32_t)T + (uint32_t)t, stream_id(PASS_T2, 0u), stage_}
                                 : RowKey{erow, stream_id(PASS_SIM, (uint32_t)t), stage_};
        __syncthreads();
        if (NWG == 1) trans_chain(a.W, bufA, bufB, w, n, q, ln, a.k0, a.k1, rk);
        else trans_chain_x<NWG>(a.W, bufA, bufB, w, n, q, ln, a.k0, a.k1, rk, kw, xch, sync, xn, tid, &bad_);
        // ---- both transitions into the trajectory core's tr rows: [pass][e * T + t][32] (mean 0..9, logvar 10..19, zero padding)
        if (tid < 64 && e0 + ne < E && a.tr && writer) {
            float4* dst = reinterpret_cast<float4*>(a.tr + ((size_t)second * E * T + (size_t)(e0 + ne) * T + t) * 32);
            dst[q] = bufA[aswz(n, q)];
            dst[4 + q] = bufA[aswz(n, 4 + q)];
        }
        // ---- reparameterise and scatter into the trajectory arrays (torchmodel.py:368-376, 382-390)
        if (tid < SIM_FE * 10) {
            const int rr = tid / 10, k = tid - rr * 10, e = e0 + rr;
            if (e < E) {
                const float* o = reinterpret_cast<const float*>(bufA);
                const float mean = o[4 * aswz(rr, k >> 2) + (k & 3)], lv = o[4 * aswz(rr, (10 + k) >> 2) + ((10 + k) & 3)];
                const float eps = a.eps_inj ? a.eps_inj[((size_t)t * E + e) * 10 + k]
                                            : normal_elem(a.k0, a.k1, global_row(a.ids, 1, e, a.row_offset), stream_id(PASS_SIM, (uint32_t)t), stage_, k);
                const float samp = eps * expf(lv * 0.5f) + mean;
                const size_t oo = ((size_t)e * T + t) * 10 + k;
                if (writer) {
                    a.s0_traj[oo] = srow[rr * 16 + k];
                    a.ps1_traj[oo] = samp; a.mean_traj[oo] = mean; a.lv_traj[oo] = lv;
                }
                srow[rr * 16 + k] = a.use_means ? mean : samp;          // each (row, k) is read and rewritten by this thread only
            }
        }
        __syncthreads();
    }
    if (NWG > 1) {
        // A timeout anywhere in the group (the sticky flag, re-read once more: a peer may have raised it behind this workgroup's last
        // gather) poisons EVERY output of the group's episodes -- trajectory arrays, transition rows, the first step's habit posterior --
        // so that G of these episodes is NaN, never a finite number computed from a partial exchange.
        if (tid == 0 && __hip_atomic_load(sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) bad_ = 1;
        __syncthreads();
        if (bad_ && writer) {
            const float qnan = __builtin_nanf("");
            for (int i = tid; i < SIM_FE * T * 10; i += 256) {
                const int e = e0 + i / (T * 10);
                if (e < E) { const size_t oo = (size_t)e0 * T * 10 + i; a.ps1_traj[oo] = qnan; a.mean_traj[oo] = qnan; a.lv_traj[oo] = qnan; a.s0_traj[oo] = qnan; }
            }
            if (a.tr)
                for (int i = tid; i < 2 * SIM_FE * T * 32; i += 256) {
                    const int pass = i / (SIM_FE * T * 32), r = i - pass * (SIM_FE * T * 32), e = e0 + r / (T * 32);
                    if (e < E) a.tr[(size_t)pass * E * T * 32 + (size_t)e0 * T * 32 + r] = qnan;
                }
            if (a.Qpi0 && tid < SIM_FE * A && e0 + tid / A < E) a.Qpi0[(size_t)e0 * A + tid] = qnan;
        }
        // the last workgroup of the group to get here re-arms the arrival counter, the exit counter and the sticky flag for the next launch
        // (all NWG have passed every exchange and read the flag by then).  A launch that never gets here -- the call failed behind the
        // kernel's enqueue -- is covered by the engine: it zeroes the words on the stream before the next split launch (sim_sync_dirty).
        if (tid == 0 && __hip_atomic_fetch_add(sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NWG - 1) {
            __hip_atomic_store(sync, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

void launch_sim_chain(const SimChainArgs& a, hipStream_t st) {
    const size_t lds = 2 * ACT_F4 * sizeof(float4) + (SIM_FE * 16 + SIM_FE * 8) * sizeof(float);
    const int groups = (a.E + SIM_FE - 1) / SIM_FE;
    if (a.xch && a.sync && groups <= SIM_MAX_SPLIT_GROUPS) hipLaunchKernelGGL(k_sim_chain<8>, dim3(groups * 8), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(k_sim_chain<1>, dim3(groups), dim3(256), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// k_head<FR>: the three 256-wide hidden layers of the decoder / encoder head (+ the encoder's 256 -> 20 output layer) for FR batch
// rows per workgroup.  Same scheme as above with 256-float activation rows (64 quads) and NB = FR / 16 batch tiles per weight
// fragment: a 16-row workgroup streams 8 flop per weight byte out of L2 (the bound of k_trans_fused), 32 rows halve that traffic,
// so large launches use FR = 32.  The first layer's B operand comes straight from the global input rows (K up to 64 * F floats).
// ---------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ int aswz64(int n, int c4) { return n * 64 + ((c4 & ~15) | ((c4 ^ n) & 15)); }

template <int NMT, int NB, bool GLB>
__device__ __forceinline__ void gemm16h(f32x4 (&acc)[NMT][NB], const float4* __restrict__ Wp, int mt0, int KC, const float4* act,
                                        const float4* const (&xr)[NB], int n, int q, unsigned ln) {
    const __amdgpu_buffer_rsrc_t wr = rsrc16(Wp);
    float4 a0[NMT], a1[NMT], a2[NMT], b0[NB], b1[NB], b2[NB];
    auto lda = [&](float4 (&A)[NMT], int k) {
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) A[mt] = wfrag16(wr, ln, (unsigned)((mt0 + mt) * KC + k) * 64u);
    };
    auto ldb = [&](float4 (&B)[NB], int k) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) B[nb] = GLB ? xr[nb][4 * k] : act[aswz64(16 * nb + n, 4 * k + q)];
    };
    // fragments two chunks ahead through three statically indexed buffers (see gemm16); consecutive MFMAs on different accumulators
#define H16_STEP(AU, AL, BU, BL, K)                                                                                          \
    {                                                                                                                        \
        const int kn_ = (K) + 2 < KC ? (K) + 2 : KC - 1;                                                                     \
        lda(AL, kn_); ldb(BL, kn_);                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        _Pragma("unroll") for (int mt = 0; mt < NMT; ++mt) _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                  \
            acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(AU[mt].x, BU[nb].x, acc[mt][nb], 0, 0, 0);                    \
        _Pragma("unroll") for (int mt = 0; mt < NMT; ++mt) _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                  \
            acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(AU[mt].y, BU[nb].y, acc[mt][nb], 0, 0, 0);                    \
        _Pragma("unroll") for (int mt = 0; mt < NMT; ++mt) _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                  \
            acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(AU[mt].z, BU[nb].z, acc[mt][nb], 0, 0, 0);                    \
        _Pragma("unroll") for (int mt = 0; mt < NMT; ++mt) _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                  \
            acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(AU[mt].w, BU[nb].w, acc[mt][nb], 0, 0, 0);                    \
    }
    lda(a0, 0); ldb(b0, 0);
    lda(a1, KC > 1 ? 1 : 0); ldb(b1, KC > 1 ? 1 : 0);
    int kc = 0;
    for (; kc + 3 <= KC; kc += 3) {
        H16_STEP(a0, a2, b0, b2, kc)
        H16_STEP(a1, a0, b1, b0, kc + 1)
        H16_STEP(a2, a1, b2, b1, kc + 2)
    }
    if (kc < KC) H16_STEP(a0, a2, b0, b2, kc)
    if (kc + 1 < KC) H16_STEP(a1, a0, b1, b0, kc + 1)
#undef H16_STEP
}

// one layer of one wave: features 16 * mt0 .. + 16 * NMT of the NB batch tiles; bias (+ ReLU + MC-dropout) -> LDS
template <int NMT, int NB, bool GLB, bool HIDDEN>
__device__ __forceinline__ void head_layer(const float4* __restrict__ Wp, const float* __restrict__ bias, int mt0, int KC, const float4* act_in,
                                           const float4* const (&xr)[NB], float4* act_out, int n, int q, unsigned ln, uint32_t k0, uint32_t k1,
                                           uint32_t tag, const RowKey (&rk)[NB]) {
    f32x4 acc[NMT][NB];
    float4 bq[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        bq[mt] = *reinterpret_cast<const float4*>(bias + 16 * (mt0 + mt) + 4 * q);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mt][nb] = (f32x4)(0.f);
    }
    gemm16h<NMT, NB, GLB>(acc, Wp, mt0, KC, act_in, xr, n, q, ln);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        uint4 rnd = make_uint4(0u, 0u, 0u, 0u);
        if (HIDDEN) rnd = noise_words(k0, k1, tag, (uint32_t)((16 * mt0) >> 7), rk[nb].row, rk[nb].stream, rk[nb].stage);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            const int f = 16 * (mt0 + mt) + 4 * q;
            float v[4] = {acc[mt][nb][0] + bq[mt].x, acc[mt][nb][1] + bq[mt].y, acc[mt][nb][2] + bq[mt].z, acc[mt][nb][3] + bq[mt].w};
            if (HIDDEN) {
                const int wsel = (f >> 5) & 3;
                const uint32_t word = wsel == 0 ? rnd.x : wsel == 1 ? rnd.y : wsel == 2 ? rnd.z : rnd.w;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ((word >> ((f + e) & 31)) & 1u) ? fmaxf(v[e], 0.0f) * 2.0f : 0.0f;
            }
            act_out[aswz64(16 * nb + n, 4 * (mt0 + mt) + q)] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

}  // namespace

template <int FR>
__global__ void __launch_bounds__(256, 2) k_head(const HeadArgs a) {
    constexpr int NB = FR / 16;
    extern __shared__ __attribute__((aligned(16))) float4 sm[];
    float4* bufA = sm;
    float4* bufB = sm + FR * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const unsigned ln = (unsigned)lane * 16u;
    const int row0 = blockIdx.x * FR;
    RowKey rk[NB];
    const float4* xr[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int m = min(row0 + 16 * nb + n, a.M - 1);               // rows past the end recompute the last row (never stored)
        const int mg = a.m0 + m;
        const int g = mg / a.rows_per_group;
        rk[nb].row = global_row(a.gm.ids, a.gm.ids_div, mg - g * a.rows_per_group, a.row_offset);
        const uint2 key = group_key(a.gm, g);
        rk[nb].stream = key.x; rk[nb].stage = key.y;
        xr[nb] = reinterpret_cast<const float4*>(a.X) + (size_t)m * (4 * a.kc0) + q;
    }
    head_layer<4, NB, true, true>(a.W.w[0], a.W.b[0], 4 * w, a.kc0, nullptr, xr, bufA, n, q, ln, a.k0, a.k1, a.tag0 + 0, rk);
    __syncthreads();
    head_layer<4, NB, false, true>(a.W.w[1], a.W.b[1], 4 * w, 16, bufA, xr, bufB, n, q, ln, a.k0, a.k1, a.tag0 + 1, rk);
    __syncthreads();
    head_layer<4, NB, false, true>(a.W.w[2], a.W.b[2], 4 * w, 16, bufB, xr, bufA, n, q, ln, a.k0, a.k1, a.tag0 + 2, rk);
    __syncthreads();
    if (a.nl == 4) {
        if (w < a.out_tiles) head_layer<1, NB, false, false>(a.W.w[3], a.W.b[3], w, 16, bufA, xr, bufB, n, q, ln, 0u, 0u, 0u, rk);
        __syncthreads();
        const int oq = 4 * a.out_tiles;                                 // output quads per row
        for (int i = tid; i < FR * oq; i += 256) {
            const int r = i / oq, c4 = i - r * oq;
            if (row0 + r < a.M) reinterpret_cast<float4*>(a.Y)[(size_t)(row0 + r) * oq + c4] = bufB[aswz64(r, c4)];
        }
    } else {
        for (int i = tid; i < FR * 64; i += 256) {
            const int r = i >> 6, c4 = i & 63;
            if (row0 + r < a.M) reinterpret_cast<float4*>(a.Y)[(size_t)(row0 + r) * 64 + c4] = bufA[aswz64(r, c4)];
        }
    }
}

void launch_head(const HeadArgs& a, hipStream_t st) {
    if (a.M >= 16384) hipLaunchKernelGGL(k_head<32>, dim3((a.M + 31) / 32), dim3(256), 2 * 32 * 64 * sizeof(float4), st, a);
    else hipLaunchKernelGGL(k_head<16>, dim3((a.M + 15) / 16), dim3(256), 2 * 16 * 64 * sizeof(float4), st, a);
}

int init_fused_kernels() {
    if (hipFuncSetAttribute((const void*)k_head<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * 64 * sizeof(float4)) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_head<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 16 * 64 * sizeof(float4)) != hipSuccess) return 1;
    const size_t lds = 2 * ACT_F4 * sizeof(float4) + (FR * 16 + FR * 8) * sizeof(float);
    if (hipFuncSetAttribute((const void*)k_trans_fused, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ACT_F4 * sizeof(float4)) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_sim_chain<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_sim_chain<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 1;
    return 0;
}

}  // namespace efe
