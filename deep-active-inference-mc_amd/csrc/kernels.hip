// gfx950 (MI355X / CDNA4) kernels of the EFE rollout engine.  See kernels.h for the layout contract.
#include "kernels.h"

namespace efe {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// Dense layer:  Y^T[co, m] = act(bias[co] + sum_ci W[co,ci] * X[m, ci])  (+ MC-dropout)
//
// One wave owns MT x NT tiles of 32x32 (features x batch rows); a workgroup is 4 independent waves
// laid out along the row axis (they share the weight fragments through L1).  Operands go straight
// from L1/L2 to VGPRs: weights are pre-packed so that lane l of (tile, k-chunk) finds its 4 k-values
// W[co = l&31][ci = 8*kc + 4*(l>>5) + s] in one 16-byte word; an activation row's same 4 ci are 16
// contiguous bytes.  MFMA step s of a chunk contracts the ci pair {8*kc + s, 8*kc + 4 + s}: lanes 0-31
// carry the first, lanes 32-63 the second (v_mfma_f32_32x32x2_f32: A[i = l&31][k = l>>5],
// B[k = l>>5][j = l&31]).  Used for the small layers (transition / habit / decoder head / encoder head);
// the 256 -> 16384 layer has its own LDS-staged kernel (decoder.hip: k_fc4).
// ---------------------------------------------------------------------------------------------
// SK = 1: the 4 waves of a workgroup own 4 different row tiles.  SK = 4 (every layer with K >= 128, whatever the batch size, so that
// the fp32 summation order never depends on how many rows are in flight): the 4 waves split K of ONE tile four ways and wave 0 adds
// the partial accumulators in wave order through LDS -- a 512 x 512 layer over 2560 rows is then 1280 workgroups of 64-MFMA chains
// (5 per CU, evenly) instead of 320 of 256-MFMA chains (1.25 per CU: the kernel took two rounds, 24 us for 9 us of MFMA work).
template <int MT, int NT, int SK>
__global__ void __launch_bounds__(256) k_dense(const GemmArgs a) {
    __shared__ float red[SK == 1 ? 1 : 3 * MT * NT * 16 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    const int KC = a.cin >> 3;
    const int mt0 = blockIdx.y * MT;
    const int row0 = (SK == 1 ? blockIdx.x * 4 + wave : blockIdx.x) * (NT * 32);
    if (row0 >= a.n_pix) return;                      // wave-uniform (workgroup-uniform when SK = 4)
    const int kcb = SK == 1 ? 0 : (wave * KC) / SK, kce = SK == 1 ? KC : ((wave + 1) * KC) / SK;   // this wave's chunk range

    bool pv[NT];
    const float* xp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int m = row0 + nt * 32 + j;
        pv[nt] = m < a.n_pix;
        const int r = (a.x_mod > 0) ? (m % a.x_mod) : m;
        xp[nt] = (pv[nt] ? a.X + (size_t)r * a.ldx : a.zeros) + 4 * h;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.0f;

    // Software-pipelined contraction: the operand fragments of chunk kc + PD are requested before the MFMAs of chunk kc issue
    // (an unpipelined loop exposes one L2 round trip per chunk: 25 us for a 512 x 512 layer whose MFMAs take 7 us).  Weights
    // come through a buffer resource (SGPR base + SGPR chunk offset + one lane VGPR), activations through the row pointer.
    constexpr int PD = 4;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wp), 0, 0x7fffffff, 0x00020000);
    const unsigned wl = (unsigned)lane * 16u;
    auto wload = [&](int mt, int kc) -> float4 {
        const uint4v v = __builtin_amdgcn_raw_buffer_load_b128(wr, wl, (unsigned)(((mt0 + mt) * KC + kc) * 64) * 16u, 0);
        return __builtin_bit_cast(float4, v);
    };
    float4 aq[PD][MT], bq[PD][NT];
#pragma unroll
    for (int p_ = 0; p_ < PD; ++p_) {
        const int kc = kcb + p_ < kce ? kcb + p_ : kce - 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aq[p_][mt] = wload(mt, kc);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bq[p_][nt] = *reinterpret_cast<const float4*>(xp[nt] + kc * 8);
    }
    for (int kc0 = kcb; kc0 < kce; kc0 += PD) {
#pragma unroll
        for (int p_ = 0; p_ < PD; ++p_) {
            const int kc = kc0 + p_;
            if (kc >= kce) break;                      // uniform
            float4 av[MT], bv[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = aq[p_][mt];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = bq[p_][nt];
            const int kn = kc + PD < kce ? kc + PD : kce - 1;      // clamped: the tail re-reads the last chunk instead of branching
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) aq[p_][mt] = wload(mt, kn);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[p_][nt] = *reinterpret_cast<const float4*>(xp[nt] + kn * 8);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].x, bv[nt].x, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].y, bv[nt].y, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].z, bv[nt].z, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].w, bv[nt].w, acc[mt][nt], 0, 0, 0);
                }
        }
    }

    if (SK > 1) {                                      // split-K: partial tiles of waves 1..3 -> LDS, wave 0 adds them in wave order
        if (wave > 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[(((wave - 1) * MT * NT + mt * NT + nt) * 16 + e) * 64 + lane] = acc[mt][nt][e];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int ws = 0; ws < 3; ++ws)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mt][nt][e] += red[((ws * MT * NT + mt * NT + nt) * 16 + e) * 64 + lane];
    }

    // ---- epilogue: C/D layout col = lane&31 (batch row), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (feature)
    // The biases are fetched up front: a global load between the stores below would wait (s_waitcnt vmcnt(0), the counter
    // retires in order) for every store issued before it -- ~2 us of write latency per bias quad, 18 us for a K = 16 layer.
    float4 bbq[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int co = (mt0 + mt) * 32 + 8 * g4 + 4 * h;
            bbq[mt][g4] = (co < a.cout) ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (!pv[nt]) continue;
        const int m = row0 + nt * 32 + j;
        const size_t yoff = (size_t)m * a.ldy;
        uint4 rnd = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        if (a.dropout) {
            const int mg = a.m0 + m;
            const int g = mg / a.rows_per_group;
            const uint32_t r = global_row(a.gm.ids, a.gm.ids_div, mg - g * a.rows_per_group, a.row_offset);
            const uint2 key = group_key(a.gm, g);
            rnd = noise_words(a.k0, a.k1, a.tag, (uint32_t)((mt0 * 32) >> 7), r, key.x, key.y);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int co = (mt0 + mt) * 32 + 8 * g4 + 4 * h;
                if (co < a.cout) {
                    const float4 bb = bbq[mt][g4];
                    float v[4] = {acc[mt][nt][4 * g4 + 0] + bb.x, acc[mt][nt][4 * g4 + 1] + bb.y,
                                  acc[mt][nt][4 * g4 + 2] + bb.z, acc[mt][nt][4 * g4 + 3] + bb.w};
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                    }
                    if (a.dropout) {
                        const uint32_t word = ((co >> 5) & 3) == 0 ? rnd.x : ((co >> 5) & 3) == 1 ? rnd.y
                                            : ((co >> 5) & 3) == 2 ? rnd.z : rnd.w;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ((word >> ((co + e) & 31)) & 1u) ? v[e] * 2.0f : 0.0f;
                    }
                    *reinterpret_cast<float4*>(a.Y + yoff + co) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

template <int MT, int NT, int SK>
static void launch_dn(const GemmArgs& a, hipStream_t st) {
    const int rows_per_wg = (SK == 1 ? 4 : 1) * NT * 32;
    dim3 grid((a.n_pix + rows_per_wg - 1) / rows_per_wg, (a.mtiles + MT - 1) / MT);
    hipLaunchKernelGGL((k_dense<MT, NT, SK>), grid, dim3(256), 0, st, a);
}

int launch_dense(int MT, int NT, const GemmArgs& a, hipStream_t st) {
    const bool sk = a.cin >= 128;         // a property of the layer only (see k_dense)
#define EFE_CASE(M_, N_) if (MT == M_ && NT == N_) { if (sk) launch_dn<M_, N_, 4>(a, st); else launch_dn<M_, N_, 1>(a, st); return 0; }
    EFE_CASE(1, 1)
    EFE_CASE(2, 1)
    EFE_CASE(1, 2)
    EFE_CASE(2, 2)
#undef EFE_CASE
    return 1;                              // no such tile shape: reported by the caller, never a process abort
}

// ---------------------------------------------------------------------------------------------
// After the transition passes of one stage: reparameterise (torchmodel.py:54-56), build the three
// decoder input groups of calculate_G (torchmodel.py:274-275, 288, 291) and the next stage's state
// (torchmodel.py:243).  Thread = (group q in 0..3S-1, row r, k in 0..15).
// ---------------------------------------------------------------------------------------------
__global__ void k_trans_post(const TransPostArgs a) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = (int)(gid & 15);
    const long qr = gid >> 4;
    const int S = a.S, R = a.R;
    if (qr >= (long)3 * S * R) return;
    const int q = (int)(qr / R);
    const int r = (int)(qr - (long)q * R);
    float out = 0.0f;
    if (k < 10) {
        int src_g; uint32_t pass, sample; bool use_mean = false;
        if (q < S)            { src_g = q;     pass = PASS_T1;  sample = q;         use_mean = a.mean_mode; }
        else if (q < 2 * S)   { src_g = q;     pass = PASS_T2;  sample = q - S;     use_mean = a.mean_mode; }
        else                  { src_g = S - 1; pass = PASS_D2B; sample = q - 2 * S; }   // LAST loop-1 sample's (mean, logvar)
        const float* tr = a.tr + ((size_t)src_g * R + r) * 32;
        const float mean = tr[k], lv = tr[10 + k];
        float eps;
        if (a.eps_inj) eps = a.eps_inj[((size_t)q * R + r) * 10 + k];
        else eps = normal_elem(a.k0, a.k1, global_row(a.ids, a.ids_div, r, a.row_offset), stream_id(pass, sample), a.stage, k);
        const float samp = eps * expf(lv * 0.5f) + mean;
        out = use_mean ? mean : samp;
        if (q == S - 1) {
            if (a.ps1_last) a.ps1_last[(size_t)r * 10 + k] = samp;
            if (a.ps1_mean_last) a.ps1_mean_last[(size_t)r * 10 + k] = mean;
            if (a.next_x) a.next_x[(size_t)r * 16 + a.pi_dim + k] = (a.carry_mean || a.mean_mode) ? mean : samp;
            if (a.given_ps1) out = a.given_ps1[(size_t)r * 10 + k];
        }
    }
    a.dec_in[qr * 16 + k] = out;
    if (q == S - 1 && a.next_x) {
        if (k < a.pi_dim) a.next_x[(size_t)r * 16 + k] = a.x[(size_t)r * 16 + k];
        if (k >= a.pi_dim + 10) a.next_x[(size_t)r * 16 + k] = 0.0f;
    }
}

void launch_trans_post(const TransPostArgs& a, hipStream_t st) {
    const long threads = (long)3 * a.S * a.R * 16;
    hipLaunchKernelGGL(k_trans_post, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a);
}

// ---------------------------------------------------------------------------------------------
// EFE term combine (torchmodel.py:278-298).  Every partial sum is formed in the
// reference's order (samples inside a stage, then stages), so fp32 rounding follows it; only the
// independent per-(stage, sample, row) pieces are evaluated in parallel (one thread per batch row
// running all D*S*22 loads back to back took 60 us).
//   phase 1: thread = (stage t, sample i, row): reward term, -entropy sum, the two image-entropy sums
//   phase 2: thread = (stage t, row): sums over the samples in order
//   phase 3: thread = row: sums over the stages in order
// ---------------------------------------------------------------------------------------------
template <int TERMS_RB>              // rows per workgroup
__global__ void __launch_bounds__(256) k_terms(const TermsArgs a) {
    extern __shared__ float sh[];                       // [D*S][RB][4] then [D][RB][6]
    const int S = a.S, R = a.R, D = a.D;
    const int r0 = blockIdx.x * TERMS_RB;
    float* p1s = sh;
    float* p2s = sh + (size_t)D * S * TERMS_RB * 4;
    const float C = 2.8378770664093453f;   // log(2*pi*e), torchutils.py:19
    for (int idx = threadIdx.x; idx < D * S * TERMS_RB; idx += blockDim.x) {
        const int rr = idx % TERMS_RB, ti = idx / TERMS_RB;
        const int t = ti / S, i = ti - t * S;
        const int r = r0 + rr;
        if (r >= R) continue;
        const size_t vb = (size_t)t * 3 * S * R;
        auto val = [&](size_t i) -> float {
            if (!a.valq) return a.val[vb + i];
            const float4 qv = reinterpret_cast<const float4*>(a.valq)[vb + i];
            return (qv.x + qv.y) + (qv.z + qv.w);
        };
        const float* tr = a.tr + (((size_t)t * 2 * S + i) * R + r) * 32 + 10;
        const float* en = a.enc + (((size_t)t * S + i) * R + r) * 32 + 10;
        float h = 0.f;
        for (int k = 0; k < 10; ++k) h += 0.5f * (C + tr[k]) + 0.5f * (C + en[k]);
        p1s[idx * 4 + 0] = a.reward_div == 0.0f ? val((size_t)i * R + r)
                                                : val((size_t)i * R + r) / a.reward_div * 10.0f;          // mean over the counted pixels * 10 (torchmodel.py:212)
        p1s[idx * 4 + 1] = -h;
        p1s[idx * 4 + 2] = val((size_t)(S + i) * R + r);
        p1s[idx * 4 + 3] = val((size_t)(2 * S + i) * R + r);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < D * TERMS_RB; idx += blockDim.x) {
        const int rr = idx % TERMS_RB, t = idx / TERMS_RB;
        float t0 = 0.f, t1 = 0.f, t21 = 0.f, t22 = 0.f;
        for (int i = 0; i < S; ++i) {
            const float* q = p1s + ((size_t)(t * S + i) * TERMS_RB + rr) * 4;
            t0 += q[0]; t1 += q[1];
        }
        t0 /= (float)S; t1 /= (float)S;
        for (int i = 0; i < S; ++i) {
            const float* q = p1s + ((size_t)(t * S + i) * TERMS_RB + rr) * 4;
            t21 += q[2]; t22 += q[3];
        }
        t21 /= (float)S; t22 /= (float)S;
        const float t2 = t21 - t22;
        float* o = p2s + (size_t)idx * 6;
        o[0] = t0; o[1] = t1; o[2] = t2; o[3] = -t0 + t1 + t2; o[4] = t21; o[5] = t22;
    }
    __syncthreads();
    if (threadIdx.x < TERMS_RB && r0 + (int)threadIdx.x < R) {
        const int rr = threadIdx.x, r = r0 + rr;
        float sG = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, p1 = 0.f, p2 = 0.f;
        for (int t = 0; t < D; ++t) {
            const float* o = p2s + ((size_t)t * TERMS_RB + rr) * 6;
            s0 += o[0]; s1 += o[1]; s2 += o[2]; sG += o[3]; p1 = o[4]; p2 = o[5];
        }
        a.G[r] = sG;
        a.terms[r] = s0; a.terms[R + r] = s1; a.terms[2 * R + r] = s2;
        if (a.t2parts) { a.t2parts[r] = p1; a.t2parts[R + r] = p2; }
    }
}

int init_small_kernels() {      // per device, from efe_create: k_terms<1> may use up to the whole LDS
    return hipFuncSetAttribute((const void*)(k_terms<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess;
}

void launch_terms(const TermsArgs& a, hipStream_t st) {
    auto bytes = [&](int rb) { return ((size_t)a.D * a.S * rb * 4 + (size_t)a.D * rb * 6) * sizeof(float); };
    if (bytes(4) <= 48 * 1024) {
        hipLaunchKernelGGL(k_terms<4>, dim3((a.R + 3) / 4), dim3(256), bytes(4), st, a);
    } else {                     // very deep / many-sample calls: one row per workgroup, LDS limit raised (D*S up to ~9000)
        hipLaunchKernelGGL(k_terms<1>, dim3(a.R), dim3(256), bytes(1), st, a);
    }
}

// ---------------------------------------------------------------------------------------------
// small plumbing kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_pack_x(const float* pi, const float* s, float* x, int R, int pi_dim, int s_dim) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = gid >> 4, k = gid & 15;
    if (r >= R) return;
    float v = 0.f;
    if (k < pi_dim) v = pi[(size_t)r * pi_dim + k];                       // torch.cat([pi, s0]) torchmodel.py:59
    else if (k < pi_dim + s_dim) v = s[(size_t)r * s_dim + (k - pi_dim)];
    x[gid] = v;
}
void launch_pack_x(const float* pi, const float* s, float* x, int R, int pi_dim, int s_dim, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_x, dim3((R * 16 + 255) / 256), dim3(256), 0, st, pi, s, x, R, pi_dim, s_dim);
}

__global__ void k_pad16(const float* s, float* x, int R, int s_dim) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = gid >> 4, k = gid & 15;
    if (r >= R) return;
    x[gid] = (k < s_dim) ? s[(size_t)r * s_dim + k] : 0.f;
}
void launch_pad16(const float* s, float* x, int R, int s_dim, hipStream_t st) {
    hipLaunchKernelGGL(k_pad16, dim3((R * 16 + 255) / 256), dim3(256), 0, st, s, x, R, s_dim);
}

// root encode -> s0 (torchmodel.py:228-234): x[r] = [pi | (use_mean ? mean : eps*exp(lv/2)+mean) | 0 0]
__global__ void k_root_post(const float* enc, const float* pi, const float* eps_inj, float* x, float* s_out, int R, int use_mean,
                            uint32_t k0, uint32_t k1, uint32_t pass, uint32_t sample, uint32_t stage, uint32_t row_offset, int pi_dim) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = gid >> 4, k = gid & 15;
    if (r >= R) return;
    float v = 0.f;
    if (k < pi_dim) v = pi ? pi[(size_t)r * pi_dim + k] : 0.f;
    else if (k < pi_dim + 10) {
        const int kk = k - pi_dim;
        const float mean = enc[(size_t)r * 32 + kk], lv = enc[(size_t)r * 32 + 10 + kk];
        if (use_mean) v = mean;
        else {
            const float eps = eps_inj ? eps_inj[(size_t)r * 10 + kk]
                                      : normal_elem(k0, k1, row_offset + r, stream_id(pass, sample), stage, kk);
            v = eps * expf(lv * 0.5f) + mean;
        }
        if (s_out) s_out[(size_t)r * 10 + kk] = v;
    }
    if (x) x[gid] = v;
}
void launch_root_post(const float* enc, const float* pi, const float* eps_inj, float* x, float* s_out, int R, int use_mean,
                      uint32_t k0, uint32_t k1, uint32_t pass, uint32_t sample, uint32_t stage, uint32_t row_offset, int pi_dim, hipStream_t st) {
    hipLaunchKernelGGL(k_root_post, dim3((R * 16 + 255) / 256), dim3(256), 0, st, enc, pi, eps_inj, x, s_out, R, use_mean,
                       k0, k1, pass, sample, stage, row_offset, pi_dim);
}

__global__ void k_split_enc(const float* enc, float* mean, float* logvar, int R) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= R * 10) return;
    const int r = gid / 10, k = gid - r * 10;
    if (mean) mean[gid] = enc[(size_t)r * 32 + k];
    if (logvar) logvar[gid] = enc[(size_t)r * 32 + 10 + k];
}
void launch_split_enc(const float* enc, float* mean, float* logvar, int R, hipStream_t st) {
    hipLaunchKernelGGL(k_split_enc, dim3((R * 10 + 255) / 256), dim3(256), 0, st, enc, mean, logvar, R);
}

// ModelTop.encode_s tail (torchmodel.py:28-30): softmax + log(q + 1e-20)
__global__ void k_softmax4(const float* l32, float* logits, float* q, float* logq, int R, int n) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float mx = -INFINITY;
    for (int k = 0; k < n; ++k) mx = fmaxf(mx, l32[(size_t)r * 32 + k]);
    float e[8], sum = 0.f;
    for (int k = 0; k < n; ++k) { e[k] = expf(l32[(size_t)r * 32 + k] - mx); sum += e[k]; }
    for (int k = 0; k < n; ++k) {
        const float qq = e[k] / sum;
        if (logits) logits[(size_t)r * n + k] = l32[(size_t)r * 32 + k];
        if (q) q[(size_t)r * n + k] = qq;
        if (logq) logq[(size_t)r * n + k] = logf(qq + 1e-20f);
    }
}
void launch_softmax4(const float* l32, float* logits, float* q, float* logq, int R, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_softmax4, dim3((R + 127) / 128), dim3(128), 0, st, l32, logits, q, logq, R, n);
}

// torch.mean(G_traj) per episode (torchmodel.py:392)
__global__ void k_mean_rows(const float* G, float* out, int E, int T) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += G[(size_t)e * T + t];
    out[e] = s / (float)T;
}
void launch_mean_rows(const float* G, float* out, int E, int T, hipStream_t st) {
    hipLaunchKernelGGL(k_mean_rows, dim3((E + 127) / 128), dim3(128), 0, st, G, out, E, T);
}

__global__ void k_fill_tr(const float* mean, const float* logvar, float* tr, int R) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = gid >> 5, k = gid & 31;
    if (r >= R) return;
    tr[gid] = (k < 10) ? mean[(size_t)r * 10 + k] : (k < 20) ? logvar[(size_t)r * 10 + (k - 10)] : 0.f;
}
void launch_fill_tr(const float* mean, const float* logvar, float* tr, int R, hipStream_t st) {
    hipLaunchKernelGGL(k_fill_tr, dim3((R * 32 + 255) / 256), dim3(256), 0, st, mean, logvar, tr, R);
}

// softmax_multi_with_log(-sum_G, n) (util.py:46-53, temperature 10; logSM is not log(SM) -- as is)
__global__ void k_posterior(const float* sumG, float* P, float* logP, int n_groups, int n, float temperature) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    float x[8], mx = -INFINITY;
    for (int k = 0; k < n; ++k) { x[k] = -sumG[(size_t)g * n + k]; mx = fmaxf(mx, x[k]); }
    float e[8], sum = 0.f;
    for (int k = 0; k < n; ++k) { x[k] -= mx; e[k] = expf(x[k] / temperature); sum += e[k]; }
    for (int k = 0; k < n; ++k) {
        P[(size_t)g * n + k] = e[k] / sum;
        logP[(size_t)g * n + k] = x[k] - logf(sum + 1e-20f);
    }
}
void launch_posterior(const float* sumG, float* P, float* logP, int n_groups, int n, float temperature, hipStream_t st) {
    hipLaunchKernelGGL(k_posterior, dim3((n_groups + 127) / 128), dim3(128), 0, st, sumG, P, logP, n_groups, n, temperature);
}

// check_reward (torchmodel.py:210-212, torchutils.py:30-37) on an arbitrary image batch: one workgroup per image.
// Same per-pixel expression as the fused decoder epilogue (target 1 for rows h < 32, 0 below; mean over pixels * 10).
__global__ void __launch_bounds__(256) k_check_reward(const float* o, float* out, int intent) {
    __shared__ float sred[4];
    const float* img = o + (size_t)blockIdx.x * 4096;
    const float D1 = 1.00001f, D0 = 0.00001f;
    float part = 0.f;
    for (int p = threadIdx.x; p < 4096; p += 256) {
        const float pr = img[p];
        part += reward_term(pr, p >> 6, p & 63, 64, 64, intent);
    }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) part += __shfl_xor(part, k);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = part;
    __syncthreads();
    // mean over the pixels that count (all 4096, or the 192 of the three bar rows) * 10
    if (threadIdx.x == 0) out[blockIdx.x] = ((sred[0] + sred[1]) + (sred[2] + sred[3])) / (intent ? 192.0f : 4096.0f) * 10.0f;
}
void launch_check_reward(const float* o, float* out, int M, int intent, hipStream_t st) {
    hipLaunchKernelGGL(k_check_reward, dim3(M), dim3(256), 0, st, o, out, intent);
}

// reparameterize (torchmodel.py:54-56 / 130-132): out = eps * exp(logvar / 2) + mean, eps from Philox or injected
__global__ void k_reparam(const float* mean, const float* logvar, const float* eps_inj, float* out, int M, int n,
                          uint32_t k0, uint32_t k1, uint32_t pass, uint32_t sample, uint32_t stage, uint32_t row_offset) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= M * n) return;
    const int r = gid / n, k = gid - r * n;
    const float eps = eps_inj ? eps_inj[gid] : normal_elem(k0, k1, row_offset + r, stream_id(pass, sample), stage, k);
    out[gid] = eps * expf(logvar[gid] * 0.5f) + mean[gid];
}
void launch_reparam(const float* mean, const float* logvar, const float* eps_inj, float* out, int M, int n, uint32_t k0, uint32_t k1,
                    uint32_t pass, uint32_t sample, uint32_t stage, uint32_t row_offset, hipStream_t st) {
    hipLaunchKernelGGL(k_reparam, dim3((M * n + 255) / 256), dim3(256), 0, st, mean, logvar, eps_inj, out, M, n, k0, k1, pass, sample,
                       stage, row_offset);
}

// ---------------------------------------------------------------------------------------------
// Dynamic-dSprites environment, batched over games: one thread = one game.
// Restates /root/reference/src/game_environment.py: tick (:113-117), up/down/left/right (:119-152),
// pi_to_action (:154-169), new_image (:84-87) with the latent resampling drawn from Philox, and
// randomize_environment_all (:72-75).  State s[7] = (colour, shape, scale, orientation, x, y, reward).
// ---------------------------------------------------------------------------------------------
enum : uint32_t { TAG_ENV = 0x60, PASS_ENV = 9 };
__device__ __forceinline__ int env_randint(uint32_t k0, uint32_t k1, uint32_t game, uint32_t stage, int latent, int size) {
    const float u = u01(noise_words(k0, k1, TAG_ENV, (uint32_t)latent, game, stream_id(PASS_ENV, 0), stage).x);
    const int v = (int)(u * (float)size);
    return v < size ? v : size - 1;
}

__global__ void k_env_step(float* state, float* last_r, const int* actions, int* round_changed, int E, int repeats,
                           uint32_t k0, uint32_t k1, uint32_t stage, uint32_t game_offset) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float* s = state + (size_t)e * 7;
    float r = last_r[e];
    const int pi = actions[e];
    bool changed = false;
    for (int i = 0; i < repeats && !changed; ++i) {
        r *= 0.95f;                                                   // tick
        if (pi == 0) {                                                // up
            s[5] += 1.0f;
            if (s[5] >= 32.0f) {
                const float x = s[4];
                if (s[1] < 0.5f) r = (x > 15.0f) ? (15.0f - x) / 16.0f : (16.0f - x) / 16.0f;      // square
                else             r = (x > 15.0f) ? (x - 15.0f) / 16.0f : (x - 16.0f) / 16.0f;      // ellipse / heart
                const int sizes[6] = {1, 3, 6, 40, 32, 32};
                for (int k = 0; k < 6; ++k) s[k] = (float)env_randint(k0, k1, game_offset + e, stage, k, sizes[k]);
                // port quirk, replicated because the oracle pins it: new_image() holds the accumulated reward as a tensor VIEW,
                // overwrites the row with a fresh sample (slot 6 = 0) and writes the view back -> the reward restarts at 0
                s[6] = 0.0f;
                changed = true;
            }
        } else if (pi == 1) { if (s[5] > 0.0f) s[5] -= 1.0f; }       // down
        else if (pi == 2) { if (s[4] < 31.0f) s[4] += 1.0f; }         // left
        else if (pi == 3) { if (s[4] > 0.0f) s[4] -= 1.0f; }          // right
    }
    last_r[e] = r;
    if (round_changed) round_changed[e] = changed ? 1 : 0;
}
void launch_env_step(float* state, float* last_r, const int* actions, int* round_changed, int E, int repeats,
                     uint32_t k0, uint32_t k1, uint32_t stage, uint32_t game_offset, hipStream_t st) {
    hipLaunchKernelGGL(k_env_step, dim3((E + 127) / 128), dim3(128), 0, st, state, last_r, actions, round_changed, E, repeats,
                       k0, k1, stage, game_offset);
}

// new_image_all (game_environment.py:83-88): fresh latents, accumulated reward (slot 6) and last_r untouched
__global__ void k_env_new_image(float* state, int E, uint32_t k0, uint32_t k1, uint32_t stage, uint32_t game_offset) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int sizes[6] = {1, 3, 6, 40, 32, 32};
    float* s = state + (size_t)e * 7;
    for (int k = 0; k < 6; ++k) s[k] = (float)env_randint(k0, k1, game_offset + e, stage, k, sizes[k]);
}
void launch_env_new_image(float* state, int E, uint32_t k0, uint32_t k1, uint32_t stage, uint32_t game_offset, hipStream_t st) {
    hipLaunchKernelGGL(k_env_new_image, dim3((E + 127) / 128), dim3(128), 0, st, state, E, k0, k1, stage, game_offset);
}

__global__ void k_env_reset(float* state, float* last_r, int E, uint32_t k0, uint32_t k1, uint32_t stage, uint32_t game_offset) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int sizes[6] = {1, 3, 6, 40, 32, 32};
    float* s = state + (size_t)e * 7;
    for (int k = 0; k < 6; ++k) s[k] = (float)env_randint(k0, k1, game_offset + e, stage, k, sizes[k]);
    const float u6 = u01(noise_words(k0, k1, TAG_ENV, 6u, game_offset + e, stream_id(PASS_ENV, 0), stage).x);
    const float u7 = u01(noise_words(k0, k1, TAG_ENV, 7u, game_offset + e, stream_id(PASS_ENV, 0), stage).x);
    // separate multiply and add: the reference rounds `rand * 20` before adding -10 (hipcc would contract to an FMA, and
    // HIP's __fmul_rn/__fadd_rn are plain operators, so the product is laundered through an empty asm)
    float t6 = u6 * 20.0f, t7 = u7 * 2.0f;
    asm volatile("" : "+v"(t6), "+v"(t7));
    s[6] = -10.0f + t6;                                               // game_environment.py:74
    last_r[e] = -1.0f + t7;                                           // :75
}
void launch_env_reset(float* state, float* last_r, int E, uint32_t k0, uint32_t k1, uint32_t stage, uint32_t game_offset, hipStream_t st) {
    hipLaunchKernelGGL(k_env_reset, dim3((E + 127) / 128), dim3(128), 0, st, state, last_r, E, k0, k1, stage, game_offset);
}

// s_to_o (game_environment.py:44-54): image lookup by the port's index rule dot(s[0:6], [1,3,6,40,32,32]) plus the
// reward bar in rows 0..2 (left half = +r, right half = -r).  One workgroup = one game; err[e] = 1 if |r| > 1.
__global__ void __launch_bounds__(256) k_env_render(const float* state, const float* last_r, const unsigned char* imgs, long n_imgs,
                                                   float* frames, int* err) {
    const int e = blockIdx.x;
    const float* s = state + (size_t)e * 7;
    const float r = last_r[e];
    long idx = (long)(s[0] * 1.0f + s[1] * 3.0f + s[2] * 6.0f + s[3] * 40.0f + s[4] * 32.0f + s[5] * 32.0f);
    if (idx < 0) idx = 0;
    if (idx >= n_imgs) idx = n_imgs - 1;
    const unsigned char* src = imgs + idx * 4096;
    float* dst = frames + (size_t)e * 4096;
    const bool pos = (r >= 0.0f && r <= 1.0f), neg = (r >= -1.0f && r < 0.0f);
    for (int p = threadIdx.x; p < 4096; p += 256) {
        float v = (float)src[p];
        const int y = p >> 6, x = p & 63;
        if (y < 3) {
            if (pos && x < 32) v = r;
            else if (neg && x >= 32) v = -r;
        }
        dst[p] = v;
    }
    if (threadIdx.x == 0 && err) err[e] = (pos || neg) ? 0 : 1;
}
void launch_env_render(const float* state, const float* last_r, const unsigned char* imgs, long n_imgs, float* frames, int* err,
                       int E, hipStream_t st) {
    hipLaunchKernelGGL(k_env_render, dim3(E), dim3(256), 0, st, state, last_r, imgs, n_imgs, frames, err);
}

}  // namespace efe
