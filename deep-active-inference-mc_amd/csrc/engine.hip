// Host side of the EFE rollout engine: context, weight packing, launch orchestration and the C ABI
// declared in include/efe_engine.h.  The schedule follows SURVEY.md section 7 ("key scheduling
// insight"): depth and the MC loops are sequential only through the tiny transition net, so one call
// runs  (1) D small transition launches,  (2) ONE batched decoder pass over rows x D x 3S evaluations,
// (3) ONE batched encoder pass over rows x D x S,  (4) a per-row term combine.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_set>
#include <vector>
#include "../../include/efe_engine.h"
#include "kernels.h"

using namespace efe;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { return ctx->fail(std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)

namespace {

constexpr int S_DIM = 10;
constexpr int64_t MAC_TRANS = 541696, MAC_DEC = 43256320, MAC_ENC = 3868960, MAC_HABIT = 18176;

enum ProfClass { PROF_MID = 0, PROF_DEC_FC = 1, PROF_DEC_FC4 = 2, PROF_CT1 = 3, PROF_CT2 = 4, PROF_CT3 = 5, PROF_FINAL = 6,
                 PROF_ENC = 7, PROF_OTHER = 8, PROF_NCLS = 9 };

struct HostTensor { std::vector<float> data; std::vector<int64_t> shape; };

struct Layer {
    float* Wp = nullptr; float* bias = nullptr;
    int cin = 0;      // padded K per tap
    int cout = 0; int mtiles = 0; int ntaps = 1;
};

struct Arena {
    std::vector<std::pair<char*, size_t>> blocks;
    size_t cur = 0, off = 0, used_total = 0;
    void reset() { cur = 0; off = 0; used_total = 0; }
};

}  // namespace

struct efe_ctx {
    int device = 0;
    std::string err;
    std::map<std::string, HostTensor> raw;
    bool committed = false;
    Layer top[3], mid[4], enc_conv[3], enc_fc[4], dec_fc[4], dec_ct[3];
    // geometry: the reference's Dynamic-dSprites configuration (pi 4, 1 x 64 x 64) runs on the fused kernels; any other
    // (pi_dim, channels, resolution) runs the generic layer-by-layer convolution path (generic.hip; SURVEY 8a-13, parity unpinned)
    int pi_dim = 4, chan = 1, res = 64;
    bool generic = false;
    bool last_s1 = false;           // resolution 32: the reference's own variant (torchmodel.py:77-80, last_strides = 1): decoder base res/2, third ConvT stride 1
    int base = 16, enc_hw[5] = {64, 31, 15, 7, 3};
    size_t img_store = 4096;        // floats per stored D1 image: C*H*W NCHW (dSprites, C = 1) or H*W*4 NHWC4 (generic)
    Layer g_fc4, g_ct[3], g_enc[4];
    float* g_wf = nullptr; float g_bf[4] = {0.f, 0.f, 0.f, 0.f};
    float* g_enc1p = nullptr;       // first encoder conv for k_conv_e: [9 taps][64 lanes][2] = W[co][h][tap], W[co][2 + h][tap]
    int64_t mac_dec = 43256320, mac_enc = 3868960, mac_trans = 541696, mac_habit = 18176;
    MlpW dec16{}, enc16{};         // decoder / encoder dense heads packed the same way (k_head)
    int enc16_kc0 = 0;
    int64_t head_unfused = 0;      // option: 1 = layer-by-layer k_dense heads (A/B experiments)
    MlpW mid16{}, top16{};         // the same transition / habit weights packed for the fused 16x16x4 kernels (fused.hip)
    int64_t mid_unfused = 0;       // option: 1 = layer-by-layer k_dense transition (A/B experiments)
    float *enc_w1 = nullptr, *enc_b1 = nullptr, *dec_wf = nullptr;
    float dec_bf = 0.f;
    float* zeros = nullptr;
    std::vector<void*> owned;      // lives as long as the context
    std::vector<void*> wbufs;      // packed weights of the current commit (freed by the next one)
    Arena arena;
    // One scratch arena per context: calls are serialised by `mu` (host threads) and ordered across streams by `done_ev`
    // (a call on a different stream than the previous one first waits for that call's last kernel), so the arena reset at the
    // start of a call never races with work still in flight.
    std::mutex mu;
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    hipEvent_t done_ev = nullptr;
    size_t high_water = 0;         // largest arena use of any call so far (bytes)
    int64_t arena_grows = 0;       // number of hipMalloc calls the arena has made
    int64_t dec_chunk = 32768, enc_chunk = 32768, poison = -1, trace = 0, dec_chunk_g = 16384;
    int64_t dec_budget_g = (int64_t)28 << 30;
    // generic decoder: images per launch group = min(dec_chunk, dec_chunk_g, dec_budget_g / activation bytes per image), so the scratch
    // block of a launch group is bounded in BYTES whatever the resolution (0.68 MB of layer activations per image at 84 x 84 on the
    // fused path, 1.6 MB with the final layer unfused); poison / trace: development only
    int64_t arena_align = 256;
    int64_t reward_intent = 0;     // option "reward_upstream_intent": 1 = the reward target the upstream NHWC code means (kernels.h reward_term), 0 = the shipped port's
    int64_t ct_fuse12 = 1;         // generic path, decoder ConvT layers 1 and 2: 1 = one kernel, layer 1's output kept in LDS (k_convt_12); 0 = one launch per layer
    int64_t enc_tiled = 2;         // generic path, encoder layers 1 and 2: 2 = one kernel, conv1 kept in LDS (k_conv_e12); 1 = LDS-tiled, one launch per layer
                                   // (k_conv_e); 0 = k_conv_g for every layer (A/B, parity of the fallbacks)
    int64_t dec_split = 1;         // dSprites path: decoder launches of <= 128 images run k_dec_b4 with four workgroups per image (0 = never: A/B)
    int64_t fuse_final_g = 1;      // generic path: last two decoder layers in one kernel (k_dec_bg); 0 = separate launches (A/B, parity tests of k_final_g)
    // OPT-IN EXPERIMENTS (bf16x3.hip): the decoder's Linear(256, 16384) and three large ConvTranspose2d layers on the 16-bit matrix pipe with
    // split operands.  mfma_bf16x3 holds the MODE: 0 = off (exact fp32, the default), 1 = three bf16 planes / six products (option
    // "mfma_bf16x3"), 2 = two fp16 planes / three products (option "mfma_f16x2").  The planes below are packed for `split_packed`.
    int64_t mfma_bf16x3 = 0;
    int split_packed = 0;
    uint16_t* fc4_b3 = nullptr;    // po_net.9's packed planes (part of wbufs)
    uint16_t* ct_b3[2] = {nullptr, nullptr};      // po_net.13 / .15 (k_dec_a's layers) as planes
    uint16_t* ct3_b3 = nullptr;                   // po_net.17 (ConvT3, k_dec_b_b3) as planes
    float s_fc4 = 1.f, s_ct[2] = {1.f, 1.f}, s_ct3 = 1.f;      // fp16 split: the power of two each layer's weights were scaled by
    int64_t b3_convt3 = 1;                        // mfma_bf16x3: ConvT3 on the bf16 pipe too (0 = the fp32 k_dec_b4 behind the two bf16 kernels, round 5's form)
    bool arch_gfx950 = false;      // hipDeviceProp_t.gcnArchName starts with gfx950 (checked at creation)
    int64_t sim_split = 1;         // simulations of <= 16 episodes: the chain kernel on eight workgroups per 8 episodes (k_sim_chain<8>); 0 = one workgroup (A/B, bit-identical)
    bool sim_sync_dirty = false;   // the last split launch's call did not complete on the host: zero sim_sync before the next one
    float* sim_xch = nullptr; int* sim_sync = nullptr;      // its exchange buffer and arrival counters / sticky timeout flag (owned; zeroed on the stream before every split launch)
    int64_t check_rows = 0;        // development: range-check efe_rows.ids on the host before every _rows call
    int64_t last_macs = 0;
    // optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline)
    unsigned prof = 0;        // bitmask of ProfClass values to time
    int cls = PROF_OTHER;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> ev_spans;
    hipEvent_t ev_get() {
        if (ev_used == ev_pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); ev_pool.push_back(e); }
        return ev_pool[ev_used++];
    }
    int trace_n = 0;
    hipEvent_t prof_begin(hipStream_t st) {
        if (trace) { (void)hipStreamSynchronize(st); fprintf(stderr, "[efe trace] launch %d class %d ...\n", ++trace_n, cls); fflush(stderr); return (hipEvent_t)1; }
        if (!(prof & (1u << cls))) return nullptr;
        hipEvent_t a = ev_get(); (void)hipEventRecord(a, st); return a;
    }
    void prof_end(hipEvent_t a, hipStream_t st) {
        if (trace) { hipError_t e = hipStreamSynchronize(st); fprintf(stderr, "[efe trace] launch %d done: %s\n", trace_n, hipGetErrorString(e)); fflush(stderr); return; }
        if (!a) return;
        hipEvent_t b = ev_get(); (void)hipEventRecord(b, st);
        ev_spans.push_back({cls, {a, b}});
    }

    std::string pending;      // error raised inside a launch helper, reported by finish()
    int fail(const std::string& m) { err = m; return 1; }

    // bump allocator over a list of device blocks; grows (synchronously) on first use at a new size
    void* alloc(size_t bytes) {
        bytes = (bytes + (size_t)arena_align - 1) / (size_t)arena_align * (size_t)arena_align;
        while (true) {
            if (arena.cur < arena.blocks.size()) {
                auto& b = arena.blocks[arena.cur];
                if (arena.off + bytes <= b.second) {
                    void* p = b.first + arena.off; arena.off += bytes; arena.used_total += bytes;
                    if (arena.used_total > high_water) high_water = arena.used_total;
                    return p;
                }
                arena.cur++; arena.off = 0;
                continue;
            }
            size_t sz = std::max(bytes, (size_t)256 << 20);
            char* p = nullptr;
            if (hipMalloc((void**)&p, sz) != hipSuccess) { err = "arena hipMalloc failed"; return nullptr; }
            arena.blocks.push_back({p, sz});
            ++arena_grows;
        }
    }
    template <class T> T* allocT(size_t n) { return reinterpret_cast<T*>(alloc(n * sizeof(T))); }
};

namespace {

// ---- weight packing into the MFMA fragment-major layout [tap][mtile][kc][lane][4] -------------------
template <class Get>
int upload_packed(efe_ctx* ctx, Layer& L, int ntaps, int cout, int cin, Get get, const float* bias_src, const int* bias_perm) {
    L.ntaps = ntaps; L.cout = cout; L.mtiles = (cout + 31) / 32; L.cin = (cin + 7) / 8 * 8;
    const int KC = L.cin / 8;
    std::vector<float> p((size_t)ntaps * L.mtiles * KC * 256);
    size_t idx = 0;
    for (int t = 0; t < ntaps; ++t)
        for (int mt = 0; mt < L.mtiles; ++mt)
            for (int kc = 0; kc < KC; ++kc)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 4; ++s) {
                        const int co = mt * 32 + (lane & 31), ci = kc * 8 + 4 * (lane >> 5) + s;
                        p[idx++] = (co < cout && ci < cin) ? get(t, co, ci) : 0.f;
                    }
    std::vector<float> b((size_t)L.mtiles * 32, 0.f);
    for (int co = 0; co < cout; ++co) b[co] = bias_src[bias_perm ? bias_perm[co] : co];
    HIPCHK(hipMalloc((void**)&L.Wp, p.size() * 4)); ctx->wbufs.push_back(L.Wp);
    HIPCHK(hipMalloc((void**)&L.bias, b.size() * 4)); ctx->wbufs.push_back(L.bias);
    HIPCHK(hipMemcpy(L.Wp, p.data(), p.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(L.bias, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    return 0;
}

const HostTensor* need(efe_ctx* ctx, const std::string& key, std::initializer_list<int64_t> shape) {
    auto it = ctx->raw.find(key);
    if (it == ctx->raw.end()) { ctx->err = "missing weight " + key; return nullptr; }
    if (it->second.shape != std::vector<int64_t>(shape)) { ctx->err = "bad shape for " + key; return nullptr; }
    return &it->second;
}

int pack_linear(efe_ctx* ctx, Layer& L, const std::string& key, int out, int in, const int* row_perm, const int* col_perm) {
    const HostTensor* w = need(ctx, key + ".weight", {out, in});
    const HostTensor* b = need(ctx, key + ".bias", {out});
    if (!w || !b) return 1;
    const float* W = w->data.data();
    return upload_packed(ctx, L, 1, out, in,
        [&](int, int co, int ci) { return W[(size_t)(row_perm ? row_perm[co] : co) * in + (col_perm ? col_perm[ci] : ci)]; },
        b->data.data(), row_perm);
}

// packed for v_mfma_f32_16x16x4_f32 (fused.hip): [16-feature tile][16-channel chunk][lane = (m, q)][s] = W[16 mt + m][16 kc + 4 q + s]
int pack_linear16(efe_ctx* ctx, const float4*& Wout, const float*& bout, const std::string& key, int out, int in, const int* col_perm = nullptr) {
    const HostTensor* w = need(ctx, key + ".weight", {out, in});
    const HostTensor* b = need(ctx, key + ".bias", {out});
    if (!w || !b) return 1;
    const int mtiles = (out + 15) / 16, KC = (in + 15) / 16;
    std::vector<float> p((size_t)mtiles * KC * 256, 0.f), bb((size_t)mtiles * 16, 0.f);
    for (int mt = 0; mt < mtiles; ++mt)
        for (int kc = 0; kc < KC; ++kc)
            for (int lane = 0; lane < 64; ++lane)
                for (int s_ = 0; s_ < 4; ++s_) {
                    const int co = 16 * mt + (lane & 15), ci = 16 * kc + 4 * (lane >> 4) + s_;
                    if (co < out && ci < in) p[(((size_t)mt * KC + kc) * 64 + lane) * 4 + s_] = w->data[(size_t)co * in + (col_perm ? col_perm[ci] : ci)];
                }
    for (int co = 0; co < out; ++co) bb[co] = b->data[co];
    float *dW = nullptr, *dB = nullptr;
    HIPCHK(hipMalloc((void**)&dW, p.size() * 4)); ctx->wbufs.push_back(dW);
    HIPCHK(hipMalloc((void**)&dB, bb.size() * 4)); ctx->wbufs.push_back(dB);
    HIPCHK(hipMemcpy(dW, p.data(), p.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dB, bb.data(), bb.size() * 4, hipMemcpyHostToDevice));
    Wout = reinterpret_cast<const float4*>(dW); bout = dB;
    return 0;
}

// ---- launch helpers ----------------------------------------------------------------------------------
struct NoiseCfg {
    uint32_t k0 = 0, k1 = 0;
    GroupMap gm{1, 1, {0, 0, 0}, 0, 0};
    int rows_per_group = 1;
    uint32_t row_offset = 0;
    const uint8_t* mask = nullptr;      // liveness of the logical rows (efe_rows.mask), entry = row / mask_div
    int mask_div = 1;
};
inline RowMask live_of(const NoiseCfg& nc, int m0) { return RowMask{nc.mask, nc.mask_div, m0, nc.rows_per_group, nc.mask ? nc.gm.ids : nullptr}; }

void fc(efe_ctx* ctx, const Layer& L, const float* X, int ldx, int x_mod, float* Y, int ldy, int M, bool relu, bool drop,
        uint32_t tag, const NoiseCfg& nc, int m0, hipStream_t st) {
    GemmArgs a{};
    a.Wp = L.Wp; a.bias = L.bias; a.X = X; a.Y = Y; a.zeros = ctx->zeros;
    a.n_pix = M; a.cin = L.cin; a.cout = L.cout; a.mtiles = L.mtiles; a.ldx = ldx; a.ldy = ldy; a.x_mod = x_mod;
    a.relu = relu; a.dropout = drop; a.tag = tag; a.k0 = nc.k0; a.k1 = nc.k1; a.gm = nc.gm;
    a.rows_per_group = nc.rows_per_group; a.row_offset = nc.row_offset; a.m0 = m0;
    hipEvent_t e0 = ctx->prof_begin(st);
    if (L.mtiles >= 64 && !(L.mtiles & 1) && L.cin == 256 && ldx == 256 && x_mod == 0 && relu && drop) {
        if (ctx->mfma_bf16x3 && ctx->fc4_b3 && &L == &ctx->dec_fc[3]) {      // opt-in experiment
            a.Wb3 = ctx->fc4_b3; a.split = (int)ctx->mfma_bf16x3; a.wb3_scale_inv = 1.0f / ctx->s_fc4;
            launch_fc4_b3(a, st);
        }
        else launch_fc4(a, st);     // Linear(256, 64 * base^2): batch tile staged in LDS
    } else {
        // tile shape by problem size: small launches (transition / habit / heads) use 32x32 wave tiles so that the
        // grid still covers the 256 CUs
        int MT = L.mtiles == 1 ? 1 : 2, NT = 2;
        const long tiles22 = (long)((L.mtiles + MT - 1) / MT) * ((M + 63) / 64);
        if (tiles22 < 2048) { NT = 1; if (tiles22 * 2 < 2048) MT = 1; }
        if (launch_dense(MT, NT, a, st)) ctx->pending = "launch_dense: unsupported tile shape";
    }
    ctx->prof_end(e0, st);
}

// decoder head (enc = false: X [M][16] -> Y [M][256]) or encoder head (X [M][16 * kc0] -> Y [M][32]) as one launch (fused.hip k_head)
void head(efe_ctx* ctx, bool enc, const float* X, float* Y, int M, const NoiseCfg& nc, int m0, hipStream_t st) {
    HeadArgs a{};
    a.W = enc ? ctx->enc16 : ctx->dec16; a.kc0 = enc ? ctx->enc16_kc0 : 1; a.nl = enc ? 4 : 3; a.out_tiles = 2;
    a.tag0 = enc ? TAG_ENC : TAG_DEC; a.X = X; a.Y = Y; a.M = M; a.k0 = nc.k0; a.k1 = nc.k1; a.gm = nc.gm;
    a.rows_per_group = nc.rows_per_group; a.row_offset = nc.row_offset; a.m0 = m0;
    hipEvent_t e0 = ctx->prof_begin(st);
    launch_head(a, st);
    ctx->prof_end(e0, st);
}

// ModelMid.ps_net over M = groups*R rows; X is [R][16], every group reads the same rows (x_mod).
int run_mid(efe_ctx* ctx, const float* X, int x_mod, int M, float* tr /*[M][32]*/, const NoiseCfg& nc, hipStream_t st) {
    if (!ctx->mid_unfused) {           // one launch for the four layers, activations in LDS (fused.hip)
        TransFusedArgs a{};
        a.W = ctx->mid16; a.X = X; a.tr = tr; a.M = M; a.x_mod = x_mod; a.k0 = nc.k0; a.k1 = nc.k1; a.gm = nc.gm;
        a.rows_per_group = nc.rows_per_group; a.row_offset = nc.row_offset; a.m0 = 0;
        ctx->cls = PROF_MID;
        hipEvent_t e0 = ctx->prof_begin(st);
        launch_trans_fused(a, st);
        ctx->prof_end(e0, st);
        ctx->cls = PROF_OTHER;
        ctx->last_macs += (int64_t)M * ctx->mac_trans;
        return 0;
    }
    float* h1 = ctx->allocT<float>((size_t)M * 512);
    float* h2 = ctx->allocT<float>((size_t)M * 512);
    if (!h1 || !h2) return 1;
    ctx->cls = PROF_MID;
    fc(ctx, ctx->mid[0], X, 16, x_mod, h1, 512, M, true, true, TAG_MID + 0, nc, 0, st);
    fc(ctx, ctx->mid[1], h1, 512, 0, h2, 512, M, true, true, TAG_MID + 1, nc, 0, st);
    fc(ctx, ctx->mid[2], h2, 512, 0, h1, 512, M, true, true, TAG_MID + 2, nc, 0, st);
    fc(ctx, ctx->mid[3], h1, 512, 0, tr, 32, M, false, false, 0, nc, 0, st);
    ctx->cls = PROF_OTHER;
    ctx->last_macs += (int64_t)M * ctx->mac_trans;
    return 0;
}

// does the generic decoder run its last two layers as k_dec_bg?  Decided from the geometry BEFORE scratch is sized (y3 exists only when
// the final layer is its own launch), so that efe_reserve / efe_rollout_scratch_bytes and the call itself always agree
static bool generic_dec_fused(const efe_ctx* ctx) {
    return ctx->fuse_final_g && !ctx->last_s1 && dec_bg_ok(2 * ctx->base, 2 * ctx->base, ctx->chan);
}
// images per launch group of the generic decoder: bounded by the chunk options and by a byte budget for the group's layer activations
int64_t generic_dec_chunk(const efe_ctx* ctx, int64_t N) {
    const int64_t B = ctx->base, H2 = 2 * B, H3 = ctx->last_s1 ? 2 * B : 4 * B;
    const bool fused = generic_dec_fused(ctx);
    const int64_t per_image = (2 * B * B * 64 + H2 * H2 * 64 + (fused ? 0 : H3 * H3 * 32)) * (int64_t)sizeof(float);
    const int64_t by_bytes = std::max<int64_t>(256, ctx->dec_budget_g / per_image);
    return std::min<int64_t>(std::min<int64_t>(std::min<int64_t>(ctx->dec_chunk, ctx->dec_chunk_g), by_bytes), N);
}

// generic geometry (generic.hip): dense head -> Linear(256, 64*B*B) -> ConvT(64,64,s1) -> ConvT(64,64,s2) -> ConvT(64,32,s2) -> final conv
int run_decoder_g(efe_ctx* ctx, const float* dec_in, int N, const NoiseCfg& nc, int reward0, int store0, float* val, float* po_store,
                  hipStream_t st) {
    const int B = ctx->base, H2 = 2 * B, H3 = ctx->last_s1 ? 2 * B : 4 * B;
    const int C = (int)generic_dec_chunk(ctx, N);
    const bool fused = generic_dec_fused(ctx);
    float* hA = ctx->allocT<float>((size_t)N * 256);
    float* hB = ctx->allocT<float>((size_t)N * 256);
    float* x4 = ctx->allocT<float>((size_t)C * B * B * 64);
    float* y1 = ctx->allocT<float>((size_t)C * B * B * 64);
    float* y2 = ctx->allocT<float>((size_t)C * H2 * H2 * 64);
    float* y3 = fused ? nullptr : ctx->allocT<float>((size_t)C * H3 * H3 * 32);
    if (!hA || !hB || !x4 || !y1 || !y2 || (!fused && !y3)) return 1;
    ctx->cls = PROF_DEC_FC;
    if (!ctx->head_unfused) head(ctx, false, dec_in, hA, N, nc, 0, st);
    else {
        fc(ctx, ctx->dec_fc[0], dec_in, 16, 0, hA, 256, N, true, true, TAG_DEC + 0, nc, 0, st);
        fc(ctx, ctx->dec_fc[1], hA, 256, 0, hB, 256, N, true, true, TAG_DEC + 1, nc, 0, st);
        fc(ctx, ctx->dec_fc[2], hB, 256, 0, hA, 256, N, true, true, TAG_DEC + 2, nc, 0, st);
    }
    int cur_m0 = 0;
    auto conv = [&](const Layer& L, const float* in, float* out, int n, int hin, int cin, int hout, int cout, int mode) {
        ConvGArgs a{};
        a.in = in; a.out = out; a.Wp = L.Wp; a.bias = L.bias; a.zeros = ctx->zeros; a.n_img = n; a.Hin = hin; a.Win = hin; a.Cin = cin;
        a.Hout = hout; a.Wout = hout; a.Cout = cout; a.mtiles = L.mtiles; a.mode = mode; a.relu = 1; a.ldo = cout;
        a.live = live_of(nc, cur_m0);
        hipEvent_t e0 = ctx->prof_begin(st);
        launch_conv_g(a, st);
        ctx->prof_end(e0, st);
    };
    for (int m0 = 0; m0 < N; m0 += C) {
        const int c = std::min(C, N - m0);
        cur_m0 = m0;
        ctx->cls = PROF_DEC_FC4;
        fc(ctx, ctx->g_fc4, hA + (size_t)m0 * 256, 256, 0, x4, B * B * 64, c, true, true, TAG_DEC + 3, nc, m0, st);
        int sep12 = 1;                                    // the first two transposed layers as one launch each
        if (ctx->ct_fuse12) {                             // ... or as one kernel, layer 1's output kept in LDS (class PROF_CT2)
            ctx->cls = PROF_CT2;
            ConvT12Args f{};
            f.in = x4; f.out = y2; f.W1p = ctx->g_ct[0].Wp; f.b1 = ctx->g_ct[0].bias; f.W2p = ctx->g_ct[1].Wp; f.b2 = ctx->g_ct[1].bias;
            f.n_img = c; f.Hin = B; f.Win = B; f.live = live_of(nc, m0);
            hipEvent_t e0 = ctx->prof_begin(st);
            sep12 = launch_convt_12(f, st);
            ctx->prof_end(sep12 ? nullptr : e0, st);
        }
        if (sep12) {
            ctx->cls = PROF_CT1;
            conv(ctx->g_ct[0], x4, y1, c, B, 64, B, 64, 1);
            ctx->cls = PROF_CT2;
            conv(ctx->g_ct[1], y1, y2, c, B, 64, H2, 64, 2);
        }
        ctx->cls = PROF_CT3;
        if (fused) {        // ConvT(64,32,s2) + ReLU + ConvT(32,C,s1) + Sigmoid + per-image sums in one kernel: y3 never exists
            DecBGArgs f{};
            f.y2 = y2; f.w3 = ctx->g_ct[2].Wp; f.b3 = ctx->g_ct[2].bias; f.w4 = ctx->g_wf; for (int i = 0; i < 4; ++i) f.b4[i] = ctx->g_bf[i];
            f.rows = c; f.m0 = m0; f.rows_per_group = nc.rows_per_group; f.Hin = H2; f.Win = H2; f.C = ctx->chan; f.gm = nc.gm;
            f.reward0 = reward0; f.store0 = store0; f.reward_intent = (int)ctx->reward_intent; f.val = val; f.po = po_store; f.live = live_of(nc, m0);
            hipEvent_t e0 = ctx->prof_begin(st);
            const int rc = launch_dec_bg(f, st);
            ctx->prof_end(e0, st);
            if (rc) return ctx->fail("fused decoder tail: unsupported geometry");      // (dec_bg_ok said yes: not reachable)
            continue;
        }
        conv(ctx->g_ct[2], y2, y3, c, H2, 64, H3, 32, ctx->last_s1 ? 1 : 2);
        ctx->cls = PROF_FINAL;
        FinalGArgs f{};
        f.y3 = y3; f.w = ctx->g_wf; for (int i = 0; i < 4; ++i) f.b[i] = ctx->g_bf[i];
        f.rows = c; f.m0 = m0; f.rows_per_group = nc.rows_per_group; f.H = H3; f.W = H3; f.C = ctx->chan; f.gm = nc.gm;
        f.reward0 = reward0; f.store0 = store0; f.reward_intent = (int)ctx->reward_intent; f.val = val; f.po = po_store; f.live = live_of(nc, m0);
        hipEvent_t e0 = ctx->prof_begin(st);
        const int frc = launch_final_g(f, st);
        ctx->prof_end(e0, st);
        if (frc) return ctx->fail("final decoder layer: unsupported geometry");
    }
    ctx->cls = PROF_OTHER;
    ctx->last_macs += (int64_t)N * ctx->mac_dec;
    return 0;
}

// generic geometry: o is NHWC4 [N][res*res][4]; four Conv2d(k3,s2,p0)+ReLU, then the dense head
int run_encoder_g(efe_ctx* ctx, const float* o8, int N, const NoiseCfg& nc, float* enc, hipStream_t st) {
    const int* hw = ctx->enc_hw;
    const int C = (int)std::min<int64_t>(std::min<int64_t>(ctx->enc_chunk, 8192), N);
    float* c1 = ctx->allocT<float>((size_t)C * hw[1] * hw[1] * 32);
    float* c2 = ctx->allocT<float>((size_t)C * hw[2] * hw[2] * 32);
    float* c3 = ctx->allocT<float>((size_t)C * hw[3] * hw[3] * 64);
    float* c4 = ctx->allocT<float>((size_t)C * hw[4] * hw[4] * 64);
    float* hA = ctx->allocT<float>((size_t)C * 256);
    float* hB = ctx->allocT<float>((size_t)C * 256);
    if (!c1 || !c2 || !c3 || !c4 || !hA || !hB) return 1;
    const int flat = hw[4] * hw[4] * 64;
    float* o8w = nullptr;
    auto conv = [&](const Layer& L, const float* in, float* out, int n, int hin, int cin, int hout, int cout) {
        ConvGArgs a{};
        a.in = in; a.out = out; a.Wp = L.Wp; a.bias = L.bias; a.zeros = ctx->zeros; a.n_img = n; a.Hin = hin; a.Win = hin; a.Cin = cin;
        a.Hout = hout; a.Wout = hout; a.Cout = cout; a.mtiles = L.mtiles; a.mode = 0; a.relu = 1; a.ldo = cout;
        hipEvent_t e0 = ctx->prof_begin(st);
        launch_conv_g(a, st);
        ctx->prof_end(e0, st);
    };
    for (int m0 = 0; m0 < N; m0 += C) {
        const int c = std::min(C, N - m0);
        ctx->cls = PROF_ENC;
        // layers 1 and 2 LDS-tiled (generic_enc.hip); k_conv_g where a geometry is outside that kernel's limits
        auto conv_e = [&](int layer, const float* in, float* out, const float* Wp, const float* bias, int hin, int hout) -> int {
            ConvEArgs e{};
            e.in = in; e.out = out; e.Wp = Wp; e.bias = bias; e.n_img = c; e.Hin = hin; e.Win = hin; e.Hout = hout; e.Wout = hout;
            e.live = live_of(nc, m0);
            hipEvent_t e0 = ctx->prof_begin(st);
            const int rc = ctx->enc_tiled ? launch_conv_e(e, layer, st) : 1;
            ctx->prof_end(rc ? nullptr : e0, st);
            return rc;
        };
        const float* o4c = o8 + (size_t)m0 * hw[0] * hw[0] * GEN_IMG_LD;
        int sep12 = 1;                                    // layers 1 and 2 as one launch each
        if (ctx->enc_tiled >= 2) {                        // ... or as one kernel, conv1's output kept in LDS
            ConvE12Args e{};
            e.in = o4c; e.out = c2; e.W1p = ctx->g_enc1p; e.b1 = ctx->g_enc[0].bias; e.W2p = ctx->g_enc[1].Wp; e.b2 = ctx->g_enc[1].bias;
            e.n_img = c; e.H0 = e.W0 = hw[0]; e.H1 = e.W1 = hw[1]; e.H2 = e.W2 = hw[2]; e.live = live_of(nc, m0);
            hipEvent_t e0 = ctx->prof_begin(st);
            sep12 = launch_conv_e12(e, st);
            ctx->prof_end(sep12 ? nullptr : e0, st);
        }
        if (sep12) {
            if (conv_e(1, o4c, c1, ctx->g_enc1p, ctx->g_enc[0].bias, hw[0], hw[1])) {
                // k_conv_g contracts 8 input channels per tap: widen the image first (this path: option enc_tiled = 0)
                if (!o8w) o8w = ctx->allocT<float>((size_t)C * hw[0] * hw[0] * 8);
                if (!o8w) return 1;
                launch_nhwc4_to_8(o4c, o8w, (long)c * hw[0] * hw[0], st);
                conv(ctx->g_enc[0], o8w, c1, c, hw[0], 8, hw[1], 32);
            }
            if (conv_e(2, c1, c2, ctx->g_enc[1].Wp, ctx->g_enc[1].bias, hw[1], hw[2])) conv(ctx->g_enc[1], c1, c2, c, hw[1], 32, hw[2], 32);
        }
        conv(ctx->g_enc[2], c2, c3, c, hw[2], 32, hw[3], 64);
        conv(ctx->g_enc[3], c3, c4, c, hw[3], 64, hw[4], 64);
        if (!ctx->head_unfused) head(ctx, true, c4, enc + (size_t)m0 * 32, c, nc, m0, st);
        else {
            fc(ctx, ctx->enc_fc[0], c4, flat, 0, hA, 256, c, true, true, TAG_ENC + 0, nc, m0, st);
            fc(ctx, ctx->enc_fc[1], hA, 256, 0, hB, 256, c, true, true, TAG_ENC + 1, nc, m0, st);
            fc(ctx, ctx->enc_fc[2], hB, 256, 0, hA, 256, c, true, true, TAG_ENC + 2, nc, m0, st);
            fc(ctx, ctx->enc_fc[3], hA, 256, 0, enc + (size_t)m0 * 32, 32, c, false, false, 0, nc, m0, st);
        }
    }
    ctx->cls = PROF_OTHER;
    ctx->last_macs += (int64_t)N * ctx->mac_enc;
    return 0;
}

// ModelDown.po_net over N rows ([group][row] batch): the three small dense layers run once over all rows, then per
// chunk: dense 256->16384 (+dropout) -> k_dec_a (two transposed convs through LDS) -> k_dec_b (third transposed conv,
// final conv, sigmoid and the per-image reduction, all on chip).
// launches of at most this many images split every image over four workgroups in k_dec_b4 (per-image sums as quarters, valq)
constexpr int DEC_SPLIT_MAX = 128;
inline bool dec_split(const efe_ctx* ctx, int N) { return !ctx->generic && ctx->dec_split && N <= DEC_SPLIT_MAX; }

int run_decoder(efe_ctx* ctx, const float* dec_in /*[N][16]*/, int N, const NoiseCfg& nc, int reward0, int store0,
                float* val /*[N], or [N][4] quarter sums when dec_split(ctx, N)*/, float* po_store, hipStream_t st) {
    if (ctx->generic) return run_decoder_g(ctx, dec_in, N, nc, reward0, store0, val, po_store, st);
    const bool split = dec_split(ctx, N);
    const int C = (int)std::min<int64_t>(ctx->dec_chunk, N);
    float* hA = ctx->allocT<float>((size_t)N * 256);
    float* hB = ctx->allocT<float>((size_t)N * 256);
    float* x4 = ctx->allocT<float>((size_t)C * 16384);
    float* y2 = ctx->allocT<float>((size_t)C * 65536);
    if (!hA || !hB || !x4 || !y2) return 1;
    ctx->cls = PROF_DEC_FC;
    if (!ctx->head_unfused) head(ctx, false, dec_in, hA, N, nc, 0, st);
    else {
        fc(ctx, ctx->dec_fc[0], dec_in, 16, 0, hA, 256, N, true, true, TAG_DEC + 0, nc, 0, st);
        fc(ctx, ctx->dec_fc[1], hA, 256, 0, hB, 256, N, true, true, TAG_DEC + 1, nc, 0, st);
        fc(ctx, ctx->dec_fc[2], hB, 256, 0, hA, 256, N, true, true, TAG_DEC + 2, nc, 0, st);
    }
    const int nchunks = (N + C - 1) / C;
    int* queues = ctx->allocT<int>((size_t)nchunks);        // one image-ticket counter per k_dec_a launch
    if (!queues) return 1;
    if (!split && hipMemsetAsync(queues, 0, (size_t)nchunks * sizeof(int), st) != hipSuccess) return ctx->fail("hipMemsetAsync failed");      // (k_dec_a_s takes no tickets)
    for (int m0 = 0; m0 < N; m0 += C) {
        const int c = std::min(C, N - m0);
        ctx->cls = PROF_DEC_FC4;
        fc(ctx, ctx->dec_fc[3], hA + (size_t)m0 * 256, 256, 0, x4, 16384, c, true, true, TAG_DEC + 3, nc, m0, st);
        ctx->cls = PROF_CT2;
        DecAArgs da{};
        da.x4 = x4; da.y2 = y2; da.w1 = ctx->dec_ct[0].Wp; da.b1 = ctx->dec_ct[0].bias; da.w2 = ctx->dec_ct[1].Wp;
        da.b2 = ctx->dec_ct[1].bias; da.rows = c; da.live = live_of(nc, m0); da.queue = queues + m0 / C; da.parts = split ? 8 : 1;
        hipEvent_t e0 = ctx->prof_begin(st);
        if (ctx->mfma_bf16x3 && ctx->ct_b3[0] && !split) {      // opt-in experiment
            da.w1b3 = ctx->ct_b3[0]; da.w2b3 = ctx->ct_b3[1]; da.split = (int)ctx->mfma_bf16x3;
            da.w1s = ctx->s_ct[0]; da.w1s_inv = 1.0f / ctx->s_ct[0]; da.w2s = ctx->s_ct[1]; da.w2s_inv = 1.0f / ctx->s_ct[1];
            launch_dec_a_b3(da, st);
        }
        else launch_dec_a(da, st);
        ctx->prof_end(e0, st);
        ctx->cls = PROF_CT3;
        DecBArgs db{};
        db.y2 = y2; db.w3 = ctx->dec_ct[2].Wp; db.b3 = ctx->dec_ct[2].bias; db.w4 = ctx->dec_wf; db.b4 = ctx->dec_bf;
        db.rows = c; db.live = live_of(nc, m0); db.m0 = m0; db.rows_per_group = nc.rows_per_group; db.gm = nc.gm; db.reward0 = reward0; db.store0 = store0;
        db.val = val; db.parts = split ? 4 : 1; db.valq = split ? val : nullptr; db.po = po_store; db.reward_intent = (int)ctx->reward_intent;
        e0 = ctx->prof_begin(st);
        if (ctx->mfma_bf16x3 && ctx->b3_convt3 && ctx->ct3_b3 && !split) {      // opt-in experiment
            db.w3b3 = ctx->ct3_b3; db.split = (int)ctx->mfma_bf16x3; db.w3s = ctx->s_ct3; db.w3s_inv = 1.0f / ctx->s_ct3;
            launch_dec_b_b3(db, st);
        }
        else launch_dec_b(db, st);
        ctx->prof_end(e0, st);
    }
    ctx->cls = PROF_OTHER;
    ctx->last_macs += (int64_t)N * ctx->mac_dec;
    return 0;
}

// ModelDown.qs_net over N rows; o is [N][4096]; out enc [N][32] (mean 0..9, logvar 10..19).
int run_encoder(efe_ctx* ctx, const float* o, int N, const NoiseCfg& nc, float* enc, hipStream_t st) {
    if (ctx->generic) return run_encoder_g(ctx, o, N, nc, enc, st);
    const int C = (int)std::min<int64_t>(ctx->enc_chunk, N);
    float* c4 = ctx->allocT<float>((size_t)C * 9 * 64);
    float* hA = ctx->allocT<float>((size_t)C * 256);
    float* hB = ctx->allocT<float>((size_t)C * 256);
    if (!c4 || !hA || !hB) return 1;
    for (int m0 = 0; m0 < N; m0 += C) {
        const int c = std::min(C, N - m0);
        ctx->cls = PROF_ENC;
        EncArgs ea{};
        ea.o = o + (size_t)m0 * 4096; ea.out = c4; ea.w1 = ctx->enc_w1; ea.b1 = ctx->enc_b1;
        ea.w2 = ctx->enc_conv[0].Wp; ea.b2 = ctx->enc_conv[0].bias; ea.w3 = ctx->enc_conv[1].Wp; ea.b3 = ctx->enc_conv[1].bias;
        ea.w4 = ctx->enc_conv[2].Wp; ea.b4 = ctx->enc_conv[2].bias; ea.rows = c; ea.live = live_of(nc, m0);
        hipEvent_t e0 = ctx->prof_begin(st);
        launch_enc_trunk(ea, st);
        ctx->prof_end(e0, st);
        if (!ctx->head_unfused) head(ctx, true, c4, enc + (size_t)m0 * 32, c, nc, m0, st);
        else {
            fc(ctx, ctx->enc_fc[0], c4, 576, 0, hA, 256, c, true, true, TAG_ENC + 0, nc, m0, st);
            fc(ctx, ctx->enc_fc[1], hA, 256, 0, hB, 256, c, true, true, TAG_ENC + 1, nc, m0, st);
            fc(ctx, ctx->enc_fc[2], hB, 256, 0, hA, 256, c, true, true, TAG_ENC + 2, nc, m0, st);
            fc(ctx, ctx->enc_fc[3], hA, 256, 0, enc + (size_t)m0 * 32, 32, c, false, false, 0, nc, m0, st);
        }
    }
    ctx->cls = PROF_OTHER;
    ctx->last_macs += (int64_t)N * ctx->mac_enc;
    return 0;
}

int run_habit(efe_ctx* ctx, const float* s16 /*[M][16]*/, int M, float* l32 /*[M][32]*/, hipStream_t st) {
    float* h1 = ctx->allocT<float>((size_t)M * 128);
    float* h2 = ctx->allocT<float>((size_t)M * 128);
    if (!h1 || !h2) return 1;
    NoiseCfg nc;
    ctx->cls = PROF_OTHER;
    fc(ctx, ctx->top[0], s16, 16, 0, h1, 128, M, true, false, 0, nc, 0, st);
    fc(ctx, ctx->top[1], h1, 128, 0, h2, 128, M, true, false, 0, nc, 0, st);
    fc(ctx, ctx->top[2], h2, 128, 0, l32, 32, M, false, false, 0, nc, 0, st);
    ctx->last_macs += (int64_t)M * ctx->mac_habit;
    return 0;
}

struct CoreIO {
    const float* x0;          // [R][16] = [pi | s0 | 0 0]
    int R, D, S, mean_mode, carry_mean;
    uint32_t k0, k1, stage0, row_offset;
    const float* eps;         // nullable, per stage [3S][R][10]
    const uint8_t* mask = nullptr; int mask_div = 1;        // liveness of the R logical rows (efe_rows.mask): entry slot = row / mask_div
    const int32_t* ids = nullptr;                           // entry slot -> entry id (efe_rows.ids): noise keys and the mask follow the id
    // trajectory mode (D == 1, S == 1): T1 is given
    const float* given_ps1 = nullptr; const float* given_mean = nullptr; const float* given_logvar = nullptr;
    float* pre_tr = nullptr;  // trajectory mode: both transition groups [2][R][32] already computed (k_sim_chain)
    float *G = nullptr, *terms = nullptr, *ps1 = nullptr, *ps1_mean = nullptr, *po1 = nullptr, *t2parts = nullptr;
};

// calculate_G for D chained stages (torchmodel.py:236-243, 270-300)
int run_core(efe_ctx* ctx, const CoreIO& io, hipStream_t st) {
    const int R = io.R, D = io.D, S = io.S;
    float* tr_all = io.pre_tr ? io.pre_tr : ctx->allocT<float>((size_t)D * 2 * S * R * 32);
    float* dec_in = ctx->allocT<float>((size_t)D * 3 * S * R * 16);
    float* xbuf = ctx->allocT<float>((size_t)2 * R * 16);
    const bool vsplit = dec_split(ctx, D * 3 * S * R);          // small decoder launch: per-image sums arrive as four quarter sums
    float* val = ctx->allocT<float>((size_t)D * 3 * S * R * (vsplit ? 4 : 1));
    float* po_store = ctx->allocT<float>((size_t)D * S * R * ctx->img_store);
    float* enc = ctx->allocT<float>((size_t)D * S * R * 32);
    float* terms_tmp = io.terms ? nullptr : ctx->allocT<float>((size_t)3 * R);
    if (!tr_all || !dec_in || !xbuf || !val || !po_store || !enc) return 1;

    const float* x = io.x0;
    for (int t = 0; t < D; ++t) {
        float* tr = tr_all + (size_t)t * 2 * S * R * 32;
        NoiseCfg nc; nc.k0 = io.k0; nc.k1 = io.k1; nc.rows_per_group = R; nc.row_offset = io.row_offset;
        if (io.pre_tr) {
            // trajectory mode behind k_sim_chain: it has written both groups
        } else if (io.given_mean) {
            // trajectory mode: group T1 is supplied, only the loop-2 transition runs
            launch_fill_tr(io.given_mean, io.given_logvar, tr, R, st);
            nc.gm = GroupMap{1, 1, {PASS_T2, 0, 0}, io.stage0 + (uint32_t)t, 0};
            nc.gm.ids = io.ids; nc.gm.ids_div = io.mask_div;
            if (run_mid(ctx, x, R, R, tr + (size_t)R * 32, nc, st)) return 1;
        } else {
            nc.gm = GroupMap{2 * S, S, {PASS_T1, PASS_T2, 0}, io.stage0 + (uint32_t)t, 0};
            nc.gm.ids = io.ids; nc.gm.ids_div = io.mask_div;
            if (run_mid(ctx, x, R, 2 * S * R, tr, nc, st)) return 1;
        }
        TransPostArgs p{};
        p.tr = tr; p.x = x; p.eps_inj = io.eps ? io.eps + (size_t)t * 3 * S * R * 10 : nullptr;
        p.given_ps1 = io.given_ps1;
        p.dec_in = dec_in + (size_t)t * 3 * S * R * 16;
        float* nx = xbuf + (size_t)(t & 1) * R * 16;
        p.next_x = (t + 1 < D) ? nx : nullptr;
        p.ps1_last = (t + 1 == D) ? io.ps1 : nullptr;
        p.ps1_mean_last = (t + 1 == D) ? io.ps1_mean : nullptr;
        p.S = S; p.R = R; p.mean_mode = io.mean_mode; p.carry_mean = io.carry_mean;
        p.k0 = io.k0; p.k1 = io.k1; p.stage = io.stage0 + t; p.row_offset = io.row_offset; p.pi_dim = ctx->pi_dim;
        p.ids = io.ids; p.ids_div = io.mask_div;
        launch_trans_post(p, st);
        x = nx;
    }
    {   // one batched decoder pass over D x 3S groups
        NoiseCfg nc; nc.k0 = io.k0; nc.k1 = io.k1; nc.rows_per_group = R; nc.row_offset = io.row_offset;
        nc.gm = GroupMap{3 * S, S, {PASS_D1, PASS_D2A, PASS_D2B}, io.stage0, 0};
        nc.gm.ids = io.ids; nc.gm.ids_div = io.mask_div;
        nc.mask = io.mask; nc.mask_div = io.mask_div;
        if (run_decoder(ctx, dec_in, D * 3 * S * R, nc, 1, 1, val, po_store, st)) return 1;
    }
    {   // one batched encoder pass over the D x S loop-1 images
        NoiseCfg nc; nc.k0 = io.k0; nc.k1 = io.k1; nc.rows_per_group = R; nc.row_offset = io.row_offset;
        nc.gm = GroupMap{S, S, {PASS_E1, 0, 0}, io.stage0, 0};
        nc.gm.ids = io.ids; nc.gm.ids_div = io.mask_div;
        nc.mask = io.mask; nc.mask_div = io.mask_div;
        if (run_encoder(ctx, po_store, D * S * R, nc, enc, st)) return 1;
    }
    TermsArgs ta{};
    ta.val = val; ta.valq = vsplit ? val : nullptr; ta.tr = tr_all; ta.enc = enc; ta.D = D; ta.S = S; ta.R = R;
    // term0 of an image = 10 * mean over the pixels that count (torchmodel.py:212: all 4096, or the 192 bar pixels of the
    // upstream-intent variant); the generic geometries use the sum form of the reference's resolution-32 branch (torchmodel.py:214)
    ta.reward_div = ctx->generic ? 0.0f : (ctx->reward_intent ? 192.0f : 4096.0f);
    ta.G = io.G; ta.terms = io.terms ? io.terms : terms_tmp; ta.t2parts = io.t2parts;
    launch_terms(ta, st);
    if (io.po1) {
        const float* last = po_store + ((size_t)(D - 1) * S + (S - 1)) * R * ctx->img_store;
        if (ctx->generic) launch_to_nchw(last, io.po1, R, ctx->res * ctx->res, ctx->chan, st);
        else if (hipMemcpyAsync(io.po1, last, (size_t)R * 4096 * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return ctx->fail("po1 copy failed");
    }
    return 0;
}

// (While `st` is being captured into a hipGraph nothing executes: the event is not recorded -- an event recorded inside a capture may only
// be waited on inside it -- and the stream bookkeeping is left as it was; whoever replays the graph orders it against other streams.)
inline bool capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
}
// Start of an engine call on stream `st`: the scratch arena is reused from its start, so work of the previous call that is still
// in flight on ANOTHER stream must finish first (same stream: stream order already guarantees it).
int check_ready(efe_ctx* ctx, hipStream_t st) {
    if (!ctx) return 1;
    if (!ctx->committed) return ctx->fail("weights not committed");
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail("hipSetDevice failed");
    if (ctx->have_last && ctx->last_stream != st && !capturing(st)) {
        if (hipStreamWaitEvent(st, ctx->done_ev, 0) != hipSuccess) return ctx->fail("hipStreamWaitEvent failed");
    }
    ctx->arena.reset();
    if (ctx->poison >= 0)                  // development: every call starts on scratch filled with this byte (reads of never-written scratch show up)
        for (auto& b : ctx->arena.blocks)
            if (hipMemsetAsync(b.first, (int)(ctx->poison & 0xff), b.second, st) != hipSuccess) return ctx->fail("poison memset failed");
    ctx->last_macs = 0;
    return 0;
}

// Every exit of an entry point that passed check_ready() -- success or failure (a bad argument, an arena hipMalloc failure, an
// unsupported geometry after some kernels were already queued) -- records done_ev behind whatever was enqueued on `st`, so that the
// next call, on any stream, orders its arena reuse behind it.
struct CallGuard {
    efe_ctx* ctx; hipStream_t st;
    ~CallGuard() {
        if (capturing(st)) return;
        if (hipEventRecord(ctx->done_ev, st) == hipSuccess) { ctx->last_stream = st; ctx->have_last = true; }
    }
};

int finish(efe_ctx* ctx, hipStream_t) {
    if (!ctx->pending.empty()) { ctx->err = ctx->pending; ctx->pending.clear(); return 1; }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ctx->fail(std::string("kernel launch: ") + hipGetErrorString(e));
    return 0;
}
int finish(efe_ctx* ctx) {          // calls that use no engine scratch (environment, tree kernels, helpers)
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ctx->fail(std::string("kernel launch: ") + hipGetErrorString(e));
    return 0;
}

// Registry of live contexts: efe_create_cfg adds, efe_destroy removes.  Every entry point checks its handle against it before touching
// the object (EFE_LOCK), so a stale or made-up efe_ctx* is an error return, not a dereference of freed memory (efe_ctx_alive exports the
// test for bindings that carry the handle as an integer: csrc/torch_ops.cpp).
std::mutex g_live_mu;
std::unordered_set<const efe_ctx*> g_live;
inline bool ctx_alive(const efe_ctx* c) {
    if (!c) return false;
    std::lock_guard<std::mutex> l(g_live_mu);
    return g_live.count(c) != 0;
}
// An entry point makes the context's device current (hipSetDevice) for its launches / allocations and puts the CALLER's device back on
// every exit: in a process that drives several GPUs a call on a context of GPU 1 must not leave the thread on GPU 1.
struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(int dev) { int cur = -1; if (hipGetDevice(&cur) == hipSuccess && cur != dev) prev = cur; }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
// frees everything a context owns (device of the context current); also the failure paths of efe_create_cfg
void release_ctx(efe_ctx* ctx) {
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->done_ev) (void)hipEventDestroy(ctx->done_ev);
    for (void* p : ctx->owned) (void)hipFree(p);
    for (void* p : ctx->wbufs) (void)hipFree(p);
    for (auto& b : ctx->arena.blocks) (void)hipFree(b.first);
    delete ctx;
}
#define EFE_LOCK(ctx) if (!ctx_alive(ctx)) return 1; std::lock_guard<std::mutex> lock_((ctx)->mu); DeviceScope dev_scope_((ctx)->device)

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

int efe_abi_version(void) { return 6; }

int efe_ctx_alive(const efe_ctx* ctx) { return ctx_alive(ctx) ? 1 : 0; }

// efe_build_id(): the digest of the sources this library was compiled from -- a generated translation unit (build.py writes it at
// link time, so an edit of one kernel file recompiles that file only)

int efe_create(efe_ctx** out, int device) { return efe_create_cfg(out, device, 10, 4, 1, 64); }

int efe_create_cfg(efe_ctx** out, int device, int s_dim, int pi_dim, int channels, int resolution) {
    if (!out) return 1;
    *out = nullptr;
    // s_dim is 10 everywhere in the reference (train.py / test_demo.py); x rows are 16 floats = [pi | s | pad]
    if (s_dim != 10 || pi_dim < 2 || pi_dim > 6 || channels < 1 || channels > 3 || resolution < 32 || resolution > 128 || resolution % 4) return 7;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return 2;
    DeviceScope dev_scope_(device);                 // the caller's current device is restored on every exit
    if (hipSetDevice(device) != hipSuccess) return 3;
    if (init_small_kernels() || init_decoder_kernels() || init_fused_kernels() || init_generic_kernels() || init_bf16x3_kernels()) return 5;       // per device: a second context on another GPU needs them too
    efe_ctx* ctx = new efe_ctx();
    ctx->device = device;
    ctx->pi_dim = pi_dim; ctx->chan = channels; ctx->res = resolution;
    ctx->generic = !(channels == 1 && resolution == 64);
    {   // the split simulation chain's cross-workgroup exchange (fused.hip: sc1 accesses, no fences) is validated on gfx950 only
        hipDeviceProp_t prop;
        ctx->arch_gfx950 = hipGetDeviceProperties(&prop, device) == hipSuccess && !strncmp(prop.gcnArchName, "gfx950", 6);
        if (!ctx->arch_gfx950) ctx->sim_split = 0;
    }
    ctx->last_s1 = resolution == 32;
    ctx->base = ctx->last_s1 ? resolution / 2 : resolution / 4;
    ctx->enc_hw[0] = resolution;
    for (int i = 1; i < 5; ++i) ctx->enc_hw[i] = (ctx->enc_hw[i - 1] - 3) / 2 + 1;      // Conv2d(k3, s2, p0), SURVEY appendix A.3
    if (ctx->enc_hw[4] < 1) { delete ctx; return 7; }
    if (ctx->generic) {
        const int64_t B = ctx->base, r = resolution;
        ctx->img_store = (size_t)r * r * GEN_IMG_LD;
        ctx->mac_dec = 10 * 256 + 2 * 256 * 256 + 256 * 64 * B * B + B * B * 9 * 64 * 64 * 2 + 4 * B * B * 9 * 64 * 32 + r * r * 9 * 32 * channels;
        // (the stride-1 third layer of the resolution-32 variant works on 2B x 2B inputs: the same 4 B^2 * 9 * 64 * 32 MACs)
        const int* hw = ctx->enc_hw;
        ctx->mac_enc = (int64_t)hw[1] * hw[1] * 9 * channels * 32 + (int64_t)hw[2] * hw[2] * 9 * 32 * 32 + (int64_t)hw[3] * hw[3] * 9 * 32 * 64
                     + (int64_t)hw[4] * hw[4] * 9 * 64 * 64 + (int64_t)hw[4] * hw[4] * 64 * 256 + 2 * 256 * 256 + 256 * 20;
    }
    ctx->mac_trans = (int64_t)(pi_dim + 10) * 512 + 2 * 512 * 512 + 512 * 20;
    ctx->mac_habit = 10 * 128 + 128 * 128 + 128 * pi_dim;
    // rows beyond the batch read their K operand values from this block (k_dense / k_conv_g): it must cover the longest contraction,
    // the encoder head's 64 * h4 * h4 inputs (3136 floats at resolution 128 -- an 8 KiB block was read past its end there)
    const size_t zeros_bytes = std::max<size_t>(8192, ((size_t)ctx->enc_hw[4] * ctx->enc_hw[4] * 64 + 64) * sizeof(float));
    // (every buffer enters ctx->owned right behind its hipMalloc and every failure path below releases through release_ctx(): nothing leaks)
    if (hipMalloc((void**)&ctx->zeros, zeros_bytes) != hipSuccess) { release_ctx(ctx); return 4; }
    ctx->owned.push_back(ctx->zeros);
    if (hipMemset(ctx->zeros, 0, zeros_bytes) != hipSuccess) { release_ctx(ctx); return 4; }
    if (hipEventCreateWithFlags(&ctx->done_ev, hipEventDisableTiming) != hipSuccess) { ctx->done_ev = nullptr; release_ctx(ctx); return 6; }
    {   // exchange buffer + counters of the split simulation chain (fused.hip k_sim_chain<8>)
        const size_t xb = (size_t)SIM_MAX_SPLIT_GROUPS * 2 * 16 * 512 * sizeof(float), sb = (size_t)SIM_MAX_SPLIT_GROUPS * 4 * sizeof(int);
        if (hipMalloc((void**)&ctx->sim_xch, xb) != hipSuccess) { release_ctx(ctx); return 4; }
        ctx->owned.push_back(ctx->sim_xch);
        if (hipMalloc((void**)&ctx->sim_sync, sb) != hipSuccess) { release_ctx(ctx); return 4; }
        ctx->owned.push_back(ctx->sim_sync);
        if (hipMemset(ctx->sim_sync, 0, sb) != hipSuccess) { release_ctx(ctx); return 4; }
    }
    { std::lock_guard<std::mutex> l(g_live_mu); g_live.insert(ctx); }
    *out = ctx;
    return 0;
}

void efe_destroy(efe_ctx* ctx) {
    {   // leaves the registry first: a second efe_destroy of the same handle, or any later call with it, is refused instead of touching freed memory
        std::lock_guard<std::mutex> l(g_live_mu);
        if (!ctx || !g_live.erase(ctx)) return;
    }
    DeviceScope dev_scope_(ctx->device);
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    release_ctx(ctx);
}

int efe_get_config(efe_ctx* ctx, int* s_dim, int* pi_dim, int* channels, int* resolution) {
    if (!ctx_alive(ctx)) return 1;
    if (s_dim) *s_dim = S_DIM;
    if (pi_dim) *pi_dim = ctx->pi_dim;
    if (channels) *channels = ctx->chan;
    if (resolution) *resolution = ctx->res;
    return 0;
}

int efe_get_device(efe_ctx* ctx, int* device, char* pci_bus_id, int pci_bus_id_len) {
    if (!ctx_alive(ctx)) return 1;
    if (device) *device = ctx->device;
    if (pci_bus_id && pci_bus_id_len > 0) {
        pci_bus_id[0] = 0;
        if (hipDeviceGetPCIBusId(pci_bus_id, pci_bus_id_len, ctx->device) != hipSuccess) return ctx->fail("efe_get_device: hipDeviceGetPCIBusId failed");
    }
    return 0;
}

const char* efe_last_error(efe_ctx* ctx) { return !ctx ? "null context" : ctx_alive(ctx) ? ctx->err.c_str() : "stale or invalid context handle"; }

int efe_set_weight(efe_ctx* ctx, const char* key, const float* data_host, const int64_t* shape, int ndim) {
    if (!key || !data_host || !shape || ndim < 1 || ndim > 4) return 1;
    EFE_LOCK(ctx);
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(data_host, data_host + n);
    ctx->raw[key] = std::move(t);
    ctx->committed = false;
    return 0;
}

static int pack_fc4_b3(efe_ctx* ctx, int mode);

int efe_set_option(efe_ctx* ctx, const char* name, int64_t value) {
    if (!name) return 1;
    EFE_LOCK(ctx);
    if (!strcmp(name, "dec_chunk")) { if (value < 1) return ctx->fail("dec_chunk < 1"); ctx->dec_chunk = value; return 0; }
    if (!strcmp(name, "reward_upstream_intent")) { ctx->reward_intent = value ? 1 : 0; return 0; }
    if (!strcmp(name, "ct_fuse12")) { ctx->ct_fuse12 = value ? 1 : 0; return 0; }
    if (!strcmp(name, "enc_tiled")) { ctx->enc_tiled = value < 0 ? 0 : value > 2 ? 2 : value; return 0; }
    if (!strcmp(name, "fuse_final_g")) { ctx->fuse_final_g = value ? 1 : 0; return 0; }
    if (!strcmp(name, "dec_split")) { ctx->dec_split = value ? 1 : 0; return 0; }
    if (!strcmp(name, "dec_budget_g")) { if (value < (1 << 20)) return ctx->fail("dec_budget_g < 1 MiB"); ctx->dec_budget_g = value; return 0; }
    if (!strcmp(name, "dec_chunk_g")) { if (value < 1) return ctx->fail("dec_chunk_g < 1"); ctx->dec_chunk_g = value; return 0; }
    if (!strcmp(name, "poison")) { ctx->poison = value; return 0; }
    if (!strcmp(name, "trace")) { ctx->trace = value; return 0; }
    if (!strcmp(name, "check_rows")) { ctx->check_rows = value ? 1 : 0; return 0; }
    if (!strcmp(name, "sim_split")) {
        // the fence-free exchange of k_sim_chain<8> relies on gfx950's sc1 = agent-scope write-through / L2-bypassing accesses: refused elsewhere
        if (value && !ctx->arch_gfx950) return ctx->fail("sim_split: validated on gfx950 only (this device is another architecture)");
        ctx->sim_split = value ? 1 : 0; return 0;
    }
    if (!strcmp(name, "mfma_bf16x3") || !strcmp(name, "mfma_f16x2")) {       // opt-in experiments; the planes are packed now if the weights are already committed
        const int mode = !value ? 0 : name[5] == 'b' ? 1 : 2;
        if (mode && ctx->generic) return ctx->fail(std::string(name) + ": the experiment covers the Dynamic-dSprites geometry only");
        // (packed from ctx->raw only while raw IS the committed set -- efe_set_weight clears `committed`, the next commit packs the planes
        // with everything else; the option is on only after every plane exists: a failure part-way leaves the experiment off, not half-enabled)
        if (mode && ctx->committed && ctx->split_packed != mode) {
            HIPCHK(hipSetDevice(ctx->device));
            HIPCHK(hipDeviceSynchronize());             // work that still reads the other mode's planes
            if (pack_fc4_b3(ctx, mode)) { ctx->mfma_bf16x3 = 0; return 1; }
        }
        ctx->mfma_bf16x3 = mode;
        return 0;
    }
    if (!strcmp(name, "b3_convt3")) { ctx->b3_convt3 = value ? 1 : 0; return 0; }
    if (!strcmp(name, "arena_align")) { if (value < 256 || (value & (value - 1))) return ctx->fail("arena_align must be a power of two >= 256"); ctx->arena_align = value; return 0; }
    if (!strcmp(name, "mid_unfused")) { ctx->mid_unfused = value; return 0; }
    if (!strcmp(name, "head_unfused")) { ctx->head_unfused = value; return 0; }
    if (!strcmp(name, "enc_chunk")) { if (value < 1) return ctx->fail("enc_chunk < 1"); ctx->enc_chunk = value; return 0; }
    return ctx->fail(std::string("unknown option ") + name);
}

// options mfma_bf16x3 (mode 1) / mfma_f16x2 (mode 2): po_net.9 (rows in NHWC order), po_net.13 / .15 / .17 as 16-bit planes for the kernels of
// bf16x3.hip.  Planes of the other mode are released first; a failure leaves no planes at all.
static void drop_split_planes(efe_ctx* ctx) {
    void* old[4] = {ctx->fc4_b3, ctx->ct_b3[0], ctx->ct_b3[1], ctx->ct3_b3};
    for (void* q : old) {
        if (!q) continue;
        ctx->wbufs.erase(std::remove(ctx->wbufs.begin(), ctx->wbufs.end(), q), ctx->wbufs.end());
        (void)hipFree(q);
    }
    ctx->fc4_b3 = nullptr; ctx->ct_b3[0] = ctx->ct_b3[1] = nullptr; ctx->ct3_b3 = nullptr; ctx->split_packed = 0;
}
static int pack_fc4_b3(efe_ctx* ctx, int mode) {
    drop_split_planes(ctx);
    const int npl = mode == 2 ? 2 : 3;
    auto upload = [&](uint16_t*& dst, const std::vector<uint16_t>& v) -> int {
        if (hipMalloc((void**)&dst, v.size() * 2) != hipSuccess) { dst = nullptr; return 1; }
        ctx->wbufs.push_back(dst);
        return hipMemcpy(dst, v.data(), v.size() * 2, hipMemcpyHostToDevice) != hipSuccess;
    };
    auto fail = [&](const char* what) { drop_split_planes(ctx); return ctx->fail(std::string("split-operand planes: ") + what); };
    const HostTensor* w = need(ctx, "down.po_net.9.weight", {16384, 256});
    if (!w) { drop_split_planes(ctx); return 1; }
    std::vector<int> rowp(16384);
    for (int p = 0; p < 256; ++p) for (int c = 0; c < 64; ++c) rowp[p * 64 + c] = c * 256 + p;
    std::vector<uint16_t> planes((size_t)16384 * 256 * npl);
    ctx->s_fc4 = pack_dense_split(mode, w->data.data(), rowp.data(), 16384, 256, planes.data());
    if (upload(ctx->fc4_b3, planes)) return fail("po_net.9");
    const char* ck[2] = {"down.po_net.13", "down.po_net.15"};
    for (int i = 0; i < 2; ++i) {
        const HostTensor* cw = need(ctx, std::string(ck[i]) + ".weight", {64, 64, 3, 3});
        if (!cw) { drop_split_planes(ctx); return 1; }
        std::vector<uint16_t> cp((size_t)9 * 64 * 64 * npl);
        ctx->s_ct[i] = pack_conv_split(mode, cw->data.data(), 64, 64, cp.data());
        if (upload(ctx->ct_b3[i], cp)) return fail(ck[i]);
    }
    {
        const HostTensor* cw = need(ctx, "down.po_net.17.weight", {64, 32, 3, 3});
        if (!cw) { drop_split_planes(ctx); return 1; }
        std::vector<uint16_t> cp((size_t)4 * 9 * npl * 64 * 8);
        ctx->s_ct3 = pack_convt3_split(mode, cw->data.data(), cp.data());
        if (upload(ctx->ct3_b3, cp)) return fail("po_net.17");
    }
    ctx->split_packed = mode;
    return 0;
}

int efe_commit_weights(efe_ctx* ctx) {
    EFE_LOCK(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    // a re-commit replaces the packed buffers of the previous one: wait for work that may still read them, then free them
    if (!ctx->wbufs.empty()) {
        HIPCHK(hipDeviceSynchronize());
        for (void* p : ctx->wbufs) (void)hipFree(p);
        ctx->wbufs.clear();
        ctx->fc4_b3 = nullptr; ctx->ct_b3[0] = ctx->ct_b3[1] = nullptr; ctx->ct3_b3 = nullptr; ctx->split_packed = 0;
    }
    ctx->committed = false;
    const int A = ctx->pi_dim;
    // habit net (torchmodel.py:19-25)
    if (pack_linear(ctx, ctx->top[0], "top.qpi_net.0", 128, 10, nullptr, nullptr)) return 1;
    if (pack_linear(ctx, ctx->top[1], "top.qpi_net.2", 128, 128, nullptr, nullptr)) return 1;
    if (pack_linear(ctx, ctx->top[2], "top.qpi_net.4", A, 128, nullptr, nullptr)) return 1;
    // transition net (torchmodel.py:41-52); input = cat[pi, s0] (torchmodel.py:59)
    if (pack_linear(ctx, ctx->mid[0], "mid.ps_net.0", 512, A + 10, nullptr, nullptr)) return 1;
    if (pack_linear(ctx, ctx->mid[1], "mid.ps_net.3", 512, 512, nullptr, nullptr)) return 1;
    if (pack_linear(ctx, ctx->mid[2], "mid.ps_net.6", 512, 512, nullptr, nullptr)) return 1;
    if (pack_linear(ctx, ctx->mid[3], "mid.ps_net.9", 20, 512, nullptr, nullptr)) return 1;
    {   // the same two nets packed for the fused kernels
        const char* mk[4] = {"mid.ps_net.0", "mid.ps_net.3", "mid.ps_net.6", "mid.ps_net.9"};
        const int mo[4] = {512, 512, 512, 20}, mi[4] = {A + 10, 512, 512, 512};
        for (int i = 0; i < 4; ++i) if (pack_linear16(ctx, ctx->mid16.w[i], ctx->mid16.b[i], mk[i], mo[i], mi[i])) return 1;
        const char* tk[3] = {"top.qpi_net.0", "top.qpi_net.2", "top.qpi_net.4"};
        const int to[3] = {128, 128, A}, ti[3] = {10, 128, 128};
        for (int i = 0; i < 3; ++i) if (pack_linear16(ctx, ctx->top16.w[i], ctx->top16.b[i], tk[i], to[i], ti[i])) return 1;
    }
    // shared dense layers of the encoder / decoder heads
    if (pack_linear(ctx, ctx->enc_fc[1], "down.qs_net.12", 256, 256, nullptr, nullptr)) return 1;
    if (pack_linear(ctx, ctx->enc_fc[2], "down.qs_net.15", 256, 256, nullptr, nullptr)) return 1;
    if (pack_linear(ctx, ctx->enc_fc[3], "down.qs_net.18", 20, 256, nullptr, nullptr)) return 1;
    if (pack_linear(ctx, ctx->dec_fc[0], "down.po_net.0", 256, 10, nullptr, nullptr)) return 1;
    if (pack_linear(ctx, ctx->dec_fc[1], "down.po_net.3", 256, 256, nullptr, nullptr)) return 1;
    if (pack_linear(ctx, ctx->dec_fc[2], "down.po_net.6", 256, 256, nullptr, nullptr)) return 1;
    {   // the heads packed for k_head (the encoder's first layer follows with the geometry below)
        const char* dk[3] = {"down.po_net.0", "down.po_net.3", "down.po_net.6"};
        const int di[3] = {10, 256, 256};
        for (int i = 0; i < 3; ++i) if (pack_linear16(ctx, ctx->dec16.w[i], ctx->dec16.b[i], dk[i], 256, di[i])) return 1;
        const char* ek[3] = {"down.qs_net.12", "down.qs_net.15", "down.qs_net.18"};
        const int eo[3] = {256, 256, 20};
        for (int i = 0; i < 3; ++i) if (pack_linear16(ctx, ctx->enc16.w[i + 1], ctx->enc16.b[i + 1], ek[i], eo[i], 256)) return 1;
    }
    const char* tk[3] = {"down.po_net.13", "down.po_net.15", "down.po_net.17"};
    const int tci[3] = {64, 64, 64}, tco[3] = {64, 64, 32};
    if (ctx->generic) {
        // ---- build-defined geometry: same layer list as torchmodel.py:84-128 with the sizes the resolution implies
        const int C = ctx->chan, B = ctx->base, F = ctx->enc_hw[4] * ctx->enc_hw[4];
        const char* ck[4] = {"down.qs_net.0", "down.qs_net.2", "down.qs_net.4", "down.qs_net.6"};
        const int cci[4] = {C, 32, 32, 64}, cco[4] = {32, 32, 64, 64};
        for (int i = 0; i < 4; ++i) {
            const HostTensor* w = need(ctx, std::string(ck[i]) + ".weight", {cco[i], cci[i], 3, 3});
            const HostTensor* b = need(ctx, std::string(ck[i]) + ".bias", {cco[i]});
            if (!w || !b) return 1;
            const float* W = w->data.data(); const int Cin = cci[i];
            if (upload_packed(ctx, ctx->g_enc[i], 9, cco[i], Cin,
                              [&](int t, int co, int ci) { return W[((size_t)co * Cin + ci) * 9 + t]; }, b->data.data(), nullptr)) return 1;
        }
        {   // layer 1 for the LDS-tiled kernel (generic_enc.hip): lane (co = lane & 31, h = lane >> 5) holds its two K operands of a tap
            const HostTensor* w = need(ctx, "down.qs_net.0.weight", {32, C, 3, 3});
            if (!w) return 1;
            std::vector<float> p1(9 * 64 * 2, 0.f);
            for (int t = 0; t < 9; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 2; ++i) {
                        const int co = lane & 31, ci = 2 * i + (lane >> 5);
                        if (ci < C) p1[(t * 64 + lane) * 2 + i] = w->data[((size_t)co * C + ci) * 9 + t];
                    }
            HIPCHK(hipMalloc((void**)&ctx->g_enc1p, p1.size() * 4)); ctx->wbufs.push_back(ctx->g_enc1p);
            HIPCHK(hipMemcpy(ctx->g_enc1p, p1.data(), p1.size() * 4, hipMemcpyHostToDevice));
        }
        {   // Flatten is channel-major c*F + p; conv4's output is NHWC p*64 + c
            std::vector<int> colp((size_t)F * 64);
            for (int p_ = 0; p_ < F; ++p_) for (int c = 0; c < 64; ++c) colp[(size_t)p_ * 64 + c] = c * F + p_;
            if (pack_linear(ctx, ctx->enc_fc[0], "down.qs_net.9", 256, F * 64, nullptr, colp.data())) return 1;
            if (pack_linear16(ctx, ctx->enc16.w[0], ctx->enc16.b[0], "down.qs_net.9", 256, F * 64, colp.data())) return 1;
            ctx->enc16_kc0 = F * 4;
        }
        {   // Unflatten(1,(64,B,B)) is channel-major c*B*B + p; emitted NHWC p*64 + c
            std::vector<int> rowp((size_t)B * B * 64);
            for (int p_ = 0; p_ < B * B; ++p_) for (int c = 0; c < 64; ++c) rowp[(size_t)p_ * 64 + c] = c * B * B + p_;
            if (pack_linear(ctx, ctx->g_fc4, "down.po_net.9", B * B * 64, 256, rowp.data(), nullptr)) return 1;
        }
        for (int i = 0; i < 3; ++i) {
            const HostTensor* w = need(ctx, std::string(tk[i]) + ".weight", {tci[i], tco[i], 3, 3});
            const HostTensor* b = need(ctx, std::string(tk[i]) + ".bias", {tco[i]});
            if (!w || !b) return 1;
            const float* W = w->data.data(); const int Cout = tco[i];
            if (upload_packed(ctx, ctx->g_ct[i], 9, Cout, tci[i],
                              [&](int t, int co, int ci) { return W[((size_t)ci * Cout + co) * 9 + t]; }, b->data.data(), nullptr)) return 1;
        }
        const HostTensor* w = need(ctx, "down.po_net.19.weight", {32, C, 3, 3});
        const HostTensor* b = need(ctx, "down.po_net.19.bias", {C});
        if (!w || !b) return 1;
        std::vector<float> wf(9 * 32 * 4, 0.f);         // [tap][ci][c padded to 4]
        for (int t = 0; t < 9; ++t) for (int ci = 0; ci < 32; ++ci) for (int c = 0; c < C; ++c) wf[(t * 32 + ci) * 4 + c] = w->data[((size_t)ci * C + c) * 9 + t];
        HIPCHK(hipMalloc((void**)&ctx->g_wf, wf.size() * 4)); ctx->wbufs.push_back(ctx->g_wf);
        HIPCHK(hipMemcpy(ctx->g_wf, wf.data(), wf.size() * 4, hipMemcpyHostToDevice));
        for (int c = 0; c < 4; ++c) ctx->g_bf[c] = c < C ? b->data[c] : 0.f;
        ctx->committed = true;
        return 0;
    }
    // ---- Dynamic-dSprites geometry (1 x 64 x 64): fused kernels.  encoder (torchmodel.py:84-104): conv1 (Cin = 1) runs on the VALU: w1[tap][co]
    {
        const HostTensor* w = need(ctx, "down.qs_net.0.weight", {32, 1, 3, 3});
        const HostTensor* b = need(ctx, "down.qs_net.0.bias", {32});
        if (!w || !b) return 1;
        std::vector<float> w1(288);
        for (int t = 0; t < 9; ++t) for (int co = 0; co < 32; ++co) w1[t * 32 + co] = w->data[co * 9 + t];
        HIPCHK(hipMalloc((void**)&ctx->enc_w1, 288 * 4)); ctx->wbufs.push_back(ctx->enc_w1);
        HIPCHK(hipMalloc((void**)&ctx->enc_b1, 32 * 4)); ctx->wbufs.push_back(ctx->enc_b1);
        HIPCHK(hipMemcpy(ctx->enc_w1, w1.data(), 288 * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->enc_b1, b->data.data(), 32 * 4, hipMemcpyHostToDevice));
    }
    const char* ck[3] = {"down.qs_net.2", "down.qs_net.4", "down.qs_net.6"};
    const int cci[3] = {32, 32, 64}, cco[3] = {32, 64, 64};
    for (int i = 0; i < 3; ++i) {
        const HostTensor* w = need(ctx, std::string(ck[i]) + ".weight", {cco[i], cci[i], 3, 3});
        const HostTensor* b = need(ctx, std::string(ck[i]) + ".bias", {cco[i]});
        if (!w || !b) return 1;
        const float* W = w->data.data(); const int Cin = cci[i];
        if (upload_packed(ctx, ctx->enc_conv[i], 9, cco[i], Cin,
                          [&](int t, int co, int ci) { return W[((size_t)co * Cin + ci) * 9 + t]; }, b->data.data(), nullptr)) return 1;
        if (i == 2) {
            // conv4 runs on v_mfma_f32_16x16x4_f32 (its 9 output pixels fill 9/16 of that tile, 9/32 of the 32-wide one): same buffer size,
            // [tap][16-channel block of Cin][16-channel tile of Cout][lane = (m, q)][s] = W[16 mt + m][16 blk + 4 q + s][tap]
            std::vector<float> p((size_t)9 * 4 * 4 * 256);
            for (int t = 0; t < 9; ++t) for (int blk = 0; blk < 4; ++blk) for (int mt = 0; mt < 4; ++mt) for (int lane = 0; lane < 64; ++lane)
                for (int s_ = 0; s_ < 4; ++s_)
                    p[((((size_t)t * 4 + blk) * 4 + mt) * 64 + lane) * 4 + s_] = W[((size_t)(16 * mt + (lane & 15)) * 64 + 16 * blk + 4 * (lane >> 4) + s_) * 9 + t];
            HIPCHK(hipMemcpy(ctx->enc_conv[2].Wp, p.data(), p.size() * 4, hipMemcpyHostToDevice));
        }
    }
    {   // Flatten is channel-major c*9 + p (torchmodel.py:93); our conv4 output is NHWC p*64 + c
        std::vector<int> colp(576);
        for (int p = 0; p < 9; ++p) for (int c = 0; c < 64; ++c) colp[p * 64 + c] = c * 9 + p;
        if (pack_linear(ctx, ctx->enc_fc[0], "down.qs_net.9", 256, 576, nullptr, colp.data())) return 1;
        if (pack_linear16(ctx, ctx->enc16.w[0], ctx->enc16.b[0], "down.qs_net.9", 256, 576, colp.data())) return 1;
        ctx->enc16_kc0 = 36;
    }
    {   // Unflatten(1,(64,16,16)) is channel-major c*256 + p (torchmodel.py:119); we emit NHWC p*64 + c directly
        std::vector<int> rowp(16384);
        for (int p = 0; p < 256; ++p) for (int c = 0; c < 64; ++c) rowp[p * 64 + c] = c * 256 + p;
        if (pack_linear(ctx, ctx->dec_fc[3], "down.po_net.9", 16384, 256, rowp.data(), nullptr)) return 1;
        if (ctx->mfma_bf16x3 && pack_fc4_b3(ctx, (int)ctx->mfma_bf16x3)) return 1;
    }
    for (int i = 0; i < 3; ++i) {   // ConvTranspose2d weights are [Cin][Cout][kh][kw]
        const HostTensor* w = need(ctx, std::string(tk[i]) + ".weight", {tci[i], tco[i], 3, 3});
        const HostTensor* b = need(ctx, std::string(tk[i]) + ".bias", {tco[i]});
        if (!w || !b) return 1;
        const float* W = w->data.data(); const int Cout = tco[i];
        if (upload_packed(ctx, ctx->dec_ct[i], 9, Cout, tci[i],
                          [&](int t, int co, int ci) { return W[((size_t)ci * Cout + co) * 9 + t]; }, b->data.data(), nullptr)) return 1;
    }
    {
        const HostTensor* w = need(ctx, "down.po_net.19.weight", {32, 1, 3, 3});
        const HostTensor* b = need(ctx, "down.po_net.19.bias", {1});
        if (!w || !b) return 1;
        std::vector<float> wf(288);
        for (int t = 0; t < 9; ++t) for (int ci = 0; ci < 32; ++ci) wf[t * 32 + ci] = w->data[ci * 9 + t];
        HIPCHK(hipMalloc((void**)&ctx->dec_wf, 288 * 4)); ctx->wbufs.push_back(ctx->dec_wf);
        HIPCHK(hipMemcpy(ctx->dec_wf, wf.data(), 288 * 4, hipMemcpyHostToDevice));
        ctx->dec_bf = b->data[0];
    }
    // the host copies stay: a caller may update a single tensor with efe_set_weight and commit again
    ctx->committed = true;
    return 0;
}

int efe_env_reset(efe_ctx* ctx, float* state, float* last_r, int E, const efe_noise* nz, void* stream) {
    EFE_LOCK(ctx);
    if (!state || !last_r || !nz || E < 1) return ctx->fail("efe_env_reset: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    launch_env_reset(state, last_r, E, (uint32_t)nz->seed, (uint32_t)(nz->seed >> 32), nz->stage, nz->row_offset, (hipStream_t)stream);
    return finish(ctx);
}

int efe_env_new_image(efe_ctx* ctx, float* state, int E, const efe_noise* nz, void* stream) {
    EFE_LOCK(ctx);
    if (!state || !nz || E < 1) return ctx->fail("efe_env_new_image: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    launch_env_new_image(state, E, (uint32_t)nz->seed, (uint32_t)(nz->seed >> 32), nz->stage, nz->row_offset, (hipStream_t)stream);
    return finish(ctx);
}

int efe_env_step(efe_ctx* ctx, float* state, float* last_r, const int32_t* actions, int E, int repeats, const efe_noise* nz,
                 int32_t* round_changed, void* stream) {
    EFE_LOCK(ctx);
    if (!state || !last_r || !actions || !nz || E < 1 || repeats < 1) return ctx->fail("efe_env_step: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    launch_env_step(state, last_r, actions, round_changed, E, repeats, (uint32_t)nz->seed, (uint32_t)(nz->seed >> 32), nz->stage,
                    nz->row_offset, (hipStream_t)stream);
    return finish(ctx);
}

int efe_env_render(efe_ctx* ctx, const float* state, const float* last_r, const uint8_t* imgs, int64_t n_imgs, float* frames,
                   int32_t* err, int E, void* stream) {
    EFE_LOCK(ctx);
    if (!state || !last_r || !imgs || !frames || n_imgs < 1 || E < 1) return ctx->fail("efe_env_render: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    launch_env_render(state, last_r, imgs, (long)n_imgs, frames, err, E, (hipStream_t)stream);
    return finish(ctx);
}

namespace {
int mcts_tree(efe_ctx* ctx, const efe_mcts_tree* t, MctsTree& o) {
    if (!t || !t->W || !t->N || !t->Qpi || !t->child || !t->S || t->E < 1 || t->cap < 1 || t->A < 1 || t->A > 8 || t->s_dim < 1)
        return ctx->fail("efe_mcts: bad tree");
    o = MctsTree{t->W, t->N, t->Qpi, t->child, t->S, t->E, t->cap, t->A, t->s_dim};
    return 0;
}
}  // namespace

int efe_mcts_select(efe_ctx* ctx, const efe_mcts_tree* tree, const uint8_t* active, float C, int use_prior, int max_depth,
                    int32_t* path_nodes, int32_t* path_act, int32_t* path_len, int32_t* leaf, float* leaf_s, float* leaf_s_rep,
                    void* stream) {
    EFE_LOCK(ctx);
    MctsTree t;
    if (mcts_tree(ctx, tree, t)) return 1;
    if (!active || !path_nodes || !path_act || !path_len || !leaf || !leaf_s || !leaf_s_rep || max_depth < 1)
        return ctx->fail("efe_mcts_select: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    launch_mcts_select(t, active, C, use_prior, max_depth, path_nodes, path_act, path_len, leaf, leaf_s, leaf_s_rep, (hipStream_t)stream);
    return finish(ctx);
}

int efe_mcts_expand(efe_ctx* ctx, const efe_mcts_tree* tree, int32_t* n_nodes, const int32_t* nodes, const uint8_t* mask, const float* G,
                    const float* ps_next, void* stream) {
    EFE_LOCK(ctx);
    MctsTree t;
    if (mcts_tree(ctx, tree, t)) return 1;
    if (!n_nodes || !nodes || !mask || !G || !ps_next) return ctx->fail("efe_mcts_expand: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    launch_mcts_expand(t, n_nodes, nodes, mask, G, ps_next, (hipStream_t)stream);
    return finish(ctx);
}

int efe_mcts_backprop(efe_ctx* ctx, const efe_mcts_tree* tree, const int32_t* path_nodes, const int32_t* path_act, const int32_t* path_len,
                      const int32_t* leaf, const uint8_t* active, const float* sims, int n_sims, const float* q0, int max_depth,
                      float* g_out, uint8_t* active_out, void* stream) {
    EFE_LOCK(ctx);
    MctsTree t;
    if (mcts_tree(ctx, tree, t)) return 1;
    if (!path_nodes || !path_act || !path_len || !leaf || !active || !sims || n_sims < 1 || !q0 || !g_out || !active_out || max_depth < 1)
        return ctx->fail("efe_mcts_backprop: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    launch_mcts_backprop(t, path_nodes, path_act, path_len, leaf, active, sims, n_sims, q0, max_depth, g_out, active_out, (hipStream_t)stream);
    return finish(ctx);
}

int efe_mcts_step(efe_ctx* ctx, const efe_mcts_tree* tree, const int32_t* prev_path_act, const int32_t* prev_path_len, const float* sims, int n_sims,
                  const float* q0, float* prev_g_out, uint8_t* prev_active_out, uint8_t* active, int32_t* stop_at, int repeat, float threshold,
                  int32_t* n_active, float C, int use_prior, int max_depth, int32_t* path_nodes, int32_t* path_act, int32_t* path_len, int32_t* leaf,
                  float* leaf_s, float* leaf_s_rep, int32_t* prev_n_nodes, const float* prev_G, const float* prev_ps_next, void* stream) {
    EFE_LOCK(ctx);
    MctsTree t;
    if (mcts_tree(ctx, tree, t)) return 1;
    const int n_exp = (prev_n_nodes != nullptr) + (prev_G != nullptr) + (prev_ps_next != nullptr);
    if (!active || !stop_at || !n_active || !path_nodes || !path_act || !path_len || !leaf || !leaf_s || !leaf_s_rep || max_depth < 1 ||
        (prev_path_len && (!prev_path_act || !sims || n_sims < 1 || !q0 || !prev_g_out || !prev_active_out)) || (n_exp != 0 && n_exp != 3))
        return ctx->fail("efe_mcts_step: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    MctsStepArgs a{prev_path_act, prev_path_len, sims, n_sims, q0, prev_g_out, prev_active_out, active, stop_at, repeat, threshold, n_active,
                   C, use_prior, max_depth, path_nodes, path_act, path_len, leaf, leaf_s, leaf_s_rep, prev_n_nodes, prev_G, prev_ps_next};
    launch_mcts_step(t, a, (hipStream_t)stream);
    return finish(ctx);
}

int efe_mcts_stop(efe_ctx* ctx, const efe_mcts_tree* tree, uint8_t* active, int32_t* stop_at, int repeat, float threshold,
                  int32_t* n_active, void* stream) {
    EFE_LOCK(ctx);
    MctsTree t;
    if (mcts_tree(ctx, tree, t)) return 1;
    if (!active || !stop_at || !n_active) return ctx->fail("efe_mcts_stop: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    launch_mcts_stop(t, active, stop_at, repeat, threshold, n_active, (hipStream_t)stream);
    return finish(ctx);
}

int64_t efe_last_call_macs(efe_ctx* ctx) { return ctx ? ctx->last_macs : 0; }

int efe_prof_enable(efe_ctx* ctx, int on) {
    EFE_LOCK(ctx);
    ctx->prof = (on < 0) ? 0xFFFFFFFFu : (unsigned)on;       // < 0 = all classes, otherwise a bitmask (bit c = class c)
    ctx->ev_used = 0;
    ctx->ev_spans.clear();
    return 0;
}

int efe_prof_classes(void) { return PROF_NCLS; }

int efe_prof_read(efe_ctx* ctx, double* ms, int64_t* launches) {
    if (!ms || !launches) return 1;
    EFE_LOCK(ctx);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipDeviceSynchronize());
    for (int i = 0; i < PROF_NCLS; ++i) { ms[i] = 0.0; launches[i] = 0; }
    for (auto& sp : ctx->ev_spans) {
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, sp.second.first, sp.second.second));
        ms[sp.first] += t; launches[sp.first] += 1;
    }
    ctx->ev_used = 0;
    ctx->ev_spans.clear();
    return 0;
}

// ---- network level -------------------------------------------------------------------------------------
int efe_transition(efe_ctx* ctx, const float* pi, const float* s0, int M, const efe_noise* nz, const float* eps,
                   float* ps1, float* mean, float* logvar, void* stream) {
    EFE_LOCK(ctx);
    hipStream_t st = (hipStream_t)stream;
    if (check_ready(ctx, st)) return 1;
    CallGuard guard_{ctx, st};
    if (!pi || !s0 || !nz || M < 1) return ctx->fail("efe_transition: bad arguments");
    float* x = ctx->allocT<float>((size_t)M * 16);
    float* tr = ctx->allocT<float>((size_t)M * 32);
    if (!x || !tr) return 1;
    launch_pack_x(pi, s0, x, M, ctx->pi_dim, S_DIM, st);
    NoiseCfg nc; nc.k0 = (uint32_t)nz->seed; nc.k1 = (uint32_t)(nz->seed >> 32); nc.rows_per_group = M; nc.row_offset = nz->row_offset;
    nc.gm = GroupMap{1, 1, {nz->pass, 0, 0}, nz->stage, nz->sample};
    if (run_mid(ctx, x, 0, M, tr, nc, st)) return 1;
    launch_split_enc(tr, mean, logvar, M, st);
    if (ps1) launch_root_post(tr, nullptr, eps, nullptr, ps1, M, 0, nc.k0, nc.k1, nz->pass, nz->sample, nz->stage, nz->row_offset, ctx->pi_dim, st);
    return finish(ctx, st);
}

int efe_decoder(efe_ctx* ctx, const float* s, int M, const efe_noise* nz, float* po, void* stream) {
    EFE_LOCK(ctx);
    hipStream_t st = (hipStream_t)stream;
    if (check_ready(ctx, st)) return 1;
    CallGuard guard_{ctx, st};
    if (!s || !nz || !po || M < 1) return ctx->fail("efe_decoder: bad arguments");
    float* x = ctx->allocT<float>((size_t)M * 16);
    float* val = ctx->allocT<float>((size_t)M * 4);        // (quarter sums when the launch is split; unused by this entry point)
    if (!x || !val) return 1;
    launch_pad16(s, x, M, S_DIM, st);
    NoiseCfg nc; nc.k0 = (uint32_t)nz->seed; nc.k1 = (uint32_t)(nz->seed >> 32); nc.rows_per_group = M; nc.row_offset = nz->row_offset;
    nc.gm = GroupMap{1, 1, {nz->pass, 0, 0}, nz->stage, nz->sample};
    if (ctx->generic) {            // the generic path stores NHWC4 images: convert to the NCHW the API returns
        float* tmp = ctx->allocT<float>((size_t)M * ctx->img_store);
        if (!tmp) return 1;
        if (run_decoder(ctx, x, M, nc, 0, 1, val, tmp, st)) return 1;
        launch_to_nchw(tmp, po, M, ctx->res * ctx->res, ctx->chan, st);
    } else if (run_decoder(ctx, x, M, nc, 0, 1, val, po, st)) return 1;
    return finish(ctx, st);
}

int efe_encoder(efe_ctx* ctx, const float* o, int M, const efe_noise* nz, const float* eps, float* s, float* mean, float* logvar, void* stream) {
    EFE_LOCK(ctx);
    hipStream_t st = (hipStream_t)stream;
    if (check_ready(ctx, st)) return 1;
    CallGuard guard_{ctx, st};
    if (!o || !nz || M < 1) return ctx->fail("efe_encoder: bad arguments");
    float* enc = ctx->allocT<float>((size_t)M * 32);
    if (!enc) return 1;
    NoiseCfg nc; nc.k0 = (uint32_t)nz->seed; nc.k1 = (uint32_t)(nz->seed >> 32); nc.rows_per_group = M; nc.row_offset = nz->row_offset;
    nc.gm = GroupMap{1, 1, {nz->pass, 0, 0}, nz->stage, nz->sample};
    if (ctx->generic) {
        float* o8 = ctx->allocT<float>((size_t)M * ctx->img_store);
        if (!o8) return 1;
        launch_to_nhwc4(o, o8, M, ctx->res * ctx->res, ctx->chan, st);
        o = o8;
    }
    if (run_encoder(ctx, o, M, nc, enc, st)) return 1;
    launch_split_enc(enc, mean, logvar, M, st);
    if (s) launch_root_post(enc, nullptr, eps, nullptr, s, M, 0, nc.k0, nc.k1, nz->pass, nz->sample, nz->stage, nz->row_offset, ctx->pi_dim, st);
    return finish(ctx, st);
}

int efe_habit(efe_ctx* ctx, const float* s, int M, float* logits, float* q, float* logq, void* stream) {
    EFE_LOCK(ctx);
    hipStream_t st = (hipStream_t)stream;
    if (check_ready(ctx, st)) return 1;
    CallGuard guard_{ctx, st};
    if (!s || M < 1) return ctx->fail("efe_habit: bad arguments");
    float* x = ctx->allocT<float>((size_t)M * 16);
    float* l32 = ctx->allocT<float>((size_t)M * 32);
    if (!x || !l32) return 1;
    launch_pad16(s, x, M, S_DIM, st);
    if (run_habit(ctx, x, M, l32, st)) return 1;
    launch_softmax4(l32, logits, q, logq, M, ctx->pi_dim, st);
    return finish(ctx, st);
}

int efe_check_reward(efe_ctx* ctx, const float* o, int M, float* out, void* stream) {
    EFE_LOCK(ctx);
    if (!o || !out || M < 1) return ctx->fail("efe_check_reward: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->generic) launch_check_reward_g(o, out, M, ctx->chan, ctx->res, ctx->res, (int)ctx->reward_intent, (hipStream_t)stream);
    else launch_check_reward(o, out, M, (int)ctx->reward_intent, (hipStream_t)stream);
    return finish(ctx);
}

int efe_reparameterize(efe_ctx* ctx, const float* mean, const float* logvar, int M, int n, const efe_noise* nz, const float* eps,
                       float* out, void* stream) {
    EFE_LOCK(ctx);
    if (!mean || !logvar || !nz || !out || M < 1 || n < 1) return ctx->fail("efe_reparameterize: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    launch_reparam(mean, logvar, eps, out, M, n, (uint32_t)nz->seed, (uint32_t)(nz->seed >> 32), nz->pass, nz->sample, nz->stage,
                   nz->row_offset, (hipStream_t)stream);
    return finish(ctx);
}

// ---- EFE level -----------------------------------------------------------------------------------------
// the row set of a call (efe_rows; NULL = every row, identity)
struct RowSet { const uint8_t* mask; const int32_t* ids; int div; };
static int row_set(efe_ctx* ctx, const efe_rows* rows, int n_rows, int fixed_div, RowSet& out, const char* who, hipStream_t st) {
    if (!rows) { out = RowSet{nullptr, nullptr, fixed_div > 0 ? fixed_div : 1}; return 0; }
    const int div = fixed_div > 0 ? fixed_div : rows->rows_per_entry;
    if (div < 1 || ((rows->mask || rows->ids) && n_rows % div != 0)) { ctx->fail((std::string(who) + ": efe_rows.rows_per_entry must divide the row count").c_str()); return 1; }
    const int n_entries = n_rows / div;
    if (rows->n_total < 0 || (rows->n_total > 0 && n_entries > rows->n_total)) {
        ctx->fail((std::string(who) + ": the call has " + std::to_string(n_entries) + " entries, efe_rows.n_total says " + std::to_string(rows->n_total)).c_str());
        return 1;
    }
    if (ctx->check_rows && rows->ids && rows->n_total > 0) {          // development option: ids range-checked on the host (synchronises)
        std::vector<int32_t> hid((size_t)n_entries);
        if (hipMemcpyAsync(hid.data(), rows->ids, (size_t)n_entries * sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            ctx->fail((std::string(who) + ": check_rows could not read efe_rows.ids").c_str()); return 1;
        }
        for (int i = 0; i < n_entries; ++i)
            if (hid[(size_t)i] < 0 || hid[(size_t)i] >= rows->n_total) {
                ctx->fail((std::string(who) + ": efe_rows.ids[" + std::to_string(i) + "] = " + std::to_string(hid[(size_t)i]) + " is outside [0, n_total = " + std::to_string(rows->n_total) + ")").c_str());
                return 1;
            }
    }
    out = RowSet{rows->mask, rows->ids, div};
    return 0;
}

int efe_calculate_g(efe_ctx* ctx, const float* s0, const float* pi0, int M, int samples, int mean_mode, const efe_noise* nz,
                    const float* eps, float* G, float* terms, float* ps1, float* ps1_mean, float* po1, float* t2parts, void* stream) {
    return efe_calculate_g_rows(ctx, s0, pi0, M, samples, mean_mode, nz, eps, nullptr, G, terms, ps1, ps1_mean, po1, t2parts, stream);
}

int efe_calculate_g_rows(efe_ctx* ctx, const float* s0, const float* pi0, int M, int samples, int mean_mode, const efe_noise* nz,
                         const float* eps, const efe_rows* rows, float* G, float* terms, float* ps1, float* ps1_mean, float* po1,
                         float* t2parts, void* stream) {
    EFE_LOCK(ctx);
    hipStream_t st = (hipStream_t)stream;
    if (check_ready(ctx, st)) return 1;
    CallGuard guard_{ctx, st};
    if (!s0 || !pi0 || !nz || !G || M < 1 || samples < 1 || samples > 65535) return ctx->fail("efe_calculate_g: bad arguments");
    float* x = ctx->allocT<float>((size_t)M * 16);
    if (!x) return 1;
    launch_pack_x(pi0, s0, x, M, ctx->pi_dim, S_DIM, st);
    CoreIO io{};
    io.x0 = x; io.R = M; io.D = 1; io.S = mean_mode ? 1 : samples; io.mean_mode = mean_mode; io.carry_mean = 0;
    io.k0 = (uint32_t)nz->seed; io.k1 = (uint32_t)(nz->seed >> 32); io.stage0 = nz->stage; io.row_offset = nz->row_offset;
    io.eps = eps; io.G = G; io.terms = terms; io.ps1 = ps1; io.ps1_mean = ps1_mean; io.po1 = po1; io.t2parts = t2parts;
    RowSet rs;
    if (row_set(ctx, rows, M, 0, rs, "efe_calculate_g_rows", st)) return 1;
    io.mask = rs.mask; io.ids = rs.ids; io.mask_div = rs.div;
    if (run_core(ctx, io, st)) return 1;
    return finish(ctx, st);
}

int efe_rollout(efe_ctx* ctx, const float* o, const float* pi, int M, int steps, int samples, int calc_mean, int per_stage_mean,
                const efe_noise* nz, const float* eps, float* sum_G, float* sum_terms, float* po1, void* stream) {
    EFE_LOCK(ctx);
    hipStream_t st = (hipStream_t)stream;
    if (check_ready(ctx, st)) return 1;
    CallGuard guard_{ctx, st};
    if (!o || !pi || !nz || !sum_G || M < 1 || steps < 1 || samples < 1 || samples > 65535) return ctx->fail("efe_rollout: bad arguments");
    const uint32_t k0 = (uint32_t)nz->seed, k1 = (uint32_t)(nz->seed >> 32);
    float* enc0 = ctx->allocT<float>((size_t)M * 32);
    float* x = ctx->allocT<float>((size_t)M * 16);
    if (!enc0 || !x) return 1;
    {   // root encode + reparameterize (torchmodel.py:228-234)
        NoiseCfg nc; nc.k0 = k0; nc.k1 = k1; nc.rows_per_group = M; nc.row_offset = nz->row_offset;
        nc.gm = GroupMap{1, 1, {PASS_ROOT, 0, 0}, nz->stage, 0};
        if (ctx->generic) {
            float* o8 = ctx->allocT<float>((size_t)M * ctx->img_store);
            if (!o8) return 1;
            launch_to_nhwc4(o, o8, M, ctx->res * ctx->res, ctx->chan, st);
            o = o8;
        }
        if (run_encoder(ctx, o, M, nc, enc0, st)) return 1;
        launch_root_post(enc0, pi, eps, x, nullptr, M, calc_mean ? 1 : 0, k0, k1, PASS_ROOT, 0, nz->stage, nz->row_offset, ctx->pi_dim, st);
    }
    const int mean_mode = (per_stage_mean && calc_mean) ? 1 : 0;
    CoreIO io{};
    io.x0 = x; io.R = M; io.D = steps; io.S = mean_mode ? 1 : samples; io.mean_mode = mean_mode; io.carry_mean = calc_mean ? 1 : 0;
    io.k0 = k0; io.k1 = k1; io.stage0 = nz->stage; io.row_offset = nz->row_offset;
    io.eps = eps ? eps + (size_t)M * 10 : nullptr;
    io.G = sum_G; io.terms = sum_terms; io.po1 = po1;
    if (run_core(ctx, io, st)) return 1;
    return finish(ctx, st);
}

static int trajectory_impl(efe_ctx* ctx, const float* s0_traj, const float* ps1_traj, const float* mean_traj, const float* lv_traj,
                           const float* pi0_traj, int T, uint32_t k0, uint32_t k1, uint32_t stage, uint32_t row_offset,
                           const float* eps, float* G, const uint8_t* mask, const int32_t* ids, int mask_div, float* pre_tr, hipStream_t st) {
    float* x = ctx->allocT<float>((size_t)T * 16);
    if (!x) return 1;
    if (!pre_tr) launch_pack_x(pi0_traj, s0_traj, x, T, ctx->pi_dim, S_DIM, st);       // (the transition input: not needed when k_sim_chain has run both transitions)
    CoreIO io{};
    io.x0 = x; io.R = T; io.D = 1; io.S = 1; io.mean_mode = 0; io.carry_mean = 0;
    io.k0 = k0; io.k1 = k1; io.stage0 = stage; io.row_offset = row_offset; io.eps = eps;
    io.given_ps1 = ps1_traj; io.given_mean = mean_traj; io.given_logvar = lv_traj;
    io.G = G; io.mask = mask; io.ids = ids; io.mask_div = mask_div; io.pre_tr = pre_tr;
    return run_core(ctx, io, st);
}

int efe_trajectory(efe_ctx* ctx, const float* s0_traj, const float* ps1_traj, const float* ps1_mean_traj, const float* ps1_logvar_traj,
                   const float* pi0_traj, int T, const efe_noise* nz, const float* eps, float* G, void* stream) {
    EFE_LOCK(ctx);
    hipStream_t st = (hipStream_t)stream;
    if (check_ready(ctx, st)) return 1;
    CallGuard guard_{ctx, st};
    if (!s0_traj || !ps1_traj || !ps1_mean_traj || !ps1_logvar_traj || !pi0_traj || !nz || !G || T < 1)
        return ctx->fail("efe_trajectory: bad arguments");
    if (trajectory_impl(ctx, s0_traj, ps1_traj, ps1_mean_traj, ps1_logvar_traj, pi0_traj, T, (uint32_t)nz->seed,
                        (uint32_t)(nz->seed >> 32), nz->stage, nz->row_offset, eps, G, nullptr, nullptr, 1, nullptr, st)) return 1;
    return finish(ctx, st);
}

int efe_simulate(efe_ctx* ctx, const float* starting_s, int E, int depth, int use_means, const efe_noise* nz,
                 const float* eps, const float* u, float* G_mean, float* pi0, float* Qpi0, void* stream) {
    return efe_simulate_rows(ctx, starting_s, E, depth, use_means, nz, eps, u, nullptr, G_mean, pi0, Qpi0, stream);
}

int efe_simulate_rows(efe_ctx* ctx, const float* starting_s, int E, int depth, int use_means, const efe_noise* nz,
                      const float* eps, const float* u, const efe_rows* rows, float* G_mean, float* pi0, float* Qpi0, void* stream) {
    EFE_LOCK(ctx);
    hipStream_t st = (hipStream_t)stream;
    if (check_ready(ctx, st)) return 1;
    CallGuard guard_{ctx, st};
    if (!starting_s || !nz || !G_mean || !pi0 || E < 1 || depth < 1 || depth > 65535) return ctx->fail("efe_simulate: bad arguments");
    const uint32_t k0 = (uint32_t)nz->seed, k1 = (uint32_t)(nz->seed >> 32);
    const int T = depth;
    RowSet rs;
    if (row_set(ctx, rows, E, 1, rs, "efe_simulate_rows", (hipStream_t)stream)) return 1;      // one episode = one entry
    float* s0t = ctx->allocT<float>((size_t)E * T * 10);
    float* ps1t = ctx->allocT<float>((size_t)E * T * 10);
    float* mt = ctx->allocT<float>((size_t)E * T * 10);
    float* lvt = ctx->allocT<float>((size_t)E * T * 10);
    float* Gt = ctx->allocT<float>((size_t)E * T);
    // the trajectory core's transition rows, written by the chain kernel -- unless the A/B option mid_unfused asks for the layer-by-layer
    // transition: then the trajectory's loop-2 transition goes through run_mid like every other one (the chain kernel keeps its own rollout)
    float* trp = ctx->allocT<float>((size_t)2 * E * T * 32);
    float* pre_tr = ctx->mid_unfused ? nullptr : trp;
    if (!s0t || !ps1t || !mt || !lvt || !Gt || !trp) return 1;
    bool split_launch = false;
    {   // the whole habit-policy rollout (depth x (encode_s, sample, transition, reparameterise)) is one launch (fused.hip)
        SimChainArgs sa{};
        sa.W = ctx->mid16; sa.H = ctx->top16; sa.s0 = starting_s; sa.E = E; sa.T = T; sa.use_means = use_means;
        sa.k0 = k0; sa.k1 = k1; sa.stage = nz->stage; sa.row_offset = nz->row_offset;
        sa.eps_inj = eps; sa.u_inj = u; sa.ids = rs.ids;
        if (ctx->sim_split && (E + SIM_FE - 1) / SIM_FE <= SIM_MAX_SPLIT_GROUPS) {
            // The split form's arrival counters and sticky timeout flag re-arm themselves at the end of a launch.  Whenever the previous
            // split call did not reach its end on the host (an error behind the kernel's enqueue) they are zeroed on the stream first --
            // not before every launch: a 32-byte hipMemsetAsync costs ~25 us of the 400 us a one-episode planner iteration takes.
            sa.xch = ctx->sim_xch; sa.sync = ctx->sim_sync;
            if (ctx->sim_sync_dirty && hipMemsetAsync(ctx->sim_sync, 0, (size_t)SIM_MAX_SPLIT_GROUPS * 4 * sizeof(int), st) != hipSuccess)
                return ctx->fail("efe_simulate: sim_sync memset failed");
            ctx->sim_sync_dirty = true;             // cleared where this call has enqueued everything and found no launch error
            split_launch = true;
        }
        sa.s0_traj = s0t; sa.ps1_traj = ps1t; sa.mean_traj = mt; sa.lv_traj = lvt; sa.pi0 = pi0; sa.Qpi0 = Qpi0; sa.pi_dim = ctx->pi_dim; sa.tr = pre_tr;
        ctx->cls = PROF_MID;
        hipEvent_t e0 = ctx->prof_begin(st);
        launch_sim_chain(sa, st);
        ctx->prof_end(e0, st);
        ctx->cls = PROF_OTHER;
        ctx->last_macs += (int64_t)E * T * (2 * ctx->mac_trans + ctx->mac_habit);      // the rollout's transition and the trajectory's loop-2 transition
    }
    if (trajectory_impl(ctx, s0t, ps1t, mt, lvt, pi0, E * T, k0, k1, nz->stage, nz->row_offset * (uint32_t)T,
                        eps ? eps + (size_t)T * E * 10 : nullptr, Gt, rs.mask, rs.ids, T, pre_tr, st)) return 1;     // trajectory row e * T + t belongs to episode slot e
    launch_mean_rows(Gt, G_mean, E, T, st);
    const int rc = finish(ctx, st);
    if (rc == 0 && split_launch) ctx->sim_sync_dirty = false;
    return rc;
}

int efe_action_posterior(efe_ctx* ctx, const float* sum_G, int n_groups, int n, float temperature, float* P, float* logP, void* stream) {
    EFE_LOCK(ctx);
    if (!sum_G || !P || !logP || n_groups < 1 || n < 1 || n > 8) return ctx->fail("efe_action_posterior: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    launch_posterior(sum_G, P, logP, n_groups, n, temperature, (hipStream_t)stream);
    return finish(ctx);
}

// ---- scratch management --------------------------------------------------------------------------------
int efe_reserve(efe_ctx* ctx, int64_t bytes) {
    EFE_LOCK(ctx);
    if (bytes < 0) return ctx->fail("efe_reserve: bytes < 0");
    HIPCHK(hipSetDevice(ctx->device));
    size_t have = 0;
    for (auto& b : ctx->arena.blocks) have += b.second;
    if (ctx->arena.blocks.size() <= 1 && have >= (size_t)bytes) return 0;
    // one block that holds everything: a bump allocation never has to skip to the next block (no fragmentation, no growth)
    HIPCHK(hipDeviceSynchronize());
    for (auto& b : ctx->arena.blocks) (void)hipFree(b.first);
    ctx->arena.blocks.clear();
    ctx->arena.reset();
    const size_t sz = std::max((size_t)bytes, have);
    char* p = nullptr;
    if (hipMalloc((void**)&p, sz) != hipSuccess) return ctx->fail("efe_reserve: hipMalloc failed");
    ctx->arena.blocks.push_back({p, sz});
    return 0;
}

int64_t efe_rollout_scratch_bytes(efe_ctx* ctx, int M, int steps, int samples) {
    // mirrors the allocations of efe_rollout (run_encoder for the root, run_core: run_mid per stage, run_decoder, run_encoder);
    // tests/test_gpu_parity.py::test_reserve_no_growth keeps it honest
    if (!ctx_alive(ctx) || M < 1 || steps < 1 || samples < 1) return 0;
    const size_t A = (size_t)ctx->arena_align;
    const int64_t dec_chunk = ctx->dec_chunk, enc_chunk = ctx->enc_chunk;
    auto al = [A](size_t b) { return (b + A - 1) / A * A; };
    if (ctx->generic) {
        const size_t R = (size_t)M, D = (size_t)steps, S = (size_t)samples, B = (size_t)ctx->base, IS = ctx->img_store;
        const int* hw = ctx->enc_hw;
        size_t t = 0;
        auto enc = [&](size_t N) {
            const size_t C = std::min<size_t>(std::min<size_t>((size_t)enc_chunk, 8192), N);
            t += al(C * hw[1] * hw[1] * 32 * 4) + al(C * hw[2] * hw[2] * 32 * 4) + al(C * hw[3] * hw[3] * 64 * 4) + al(C * hw[4] * hw[4] * 64 * 4) + 2 * al(C * 256 * 4);
        };
        t += al(R * 32 * 4) + al(R * 16 * 4) + al(R * IS * 4);
        enc(R);
        t += al(D * 2 * S * R * 32 * 4) + al(D * 3 * S * R * 16 * 4) + al(2 * R * 16 * 4) + al(D * 3 * S * R * 4) + al(D * S * R * IS * 4)
           + al(D * S * R * 32 * 4) + al(3 * R * 4);
        {   const size_t N = D * 3 * S * R, C = (size_t)generic_dec_chunk(ctx, (int64_t)N);
            const bool fused = generic_dec_fused(ctx);
            t += 2 * al(N * 256 * 4) + 2 * al(C * B * B * 64 * 4) + al(C * 4 * B * B * 64 * 4) + (fused ? 0 : al(C * (size_t)ctx->res * ctx->res * 32 * 4)); }
        enc(D * S * R);
        return (int64_t)(t + ((size_t)1 << 20));
    }
    const size_t R = (size_t)M, D = (size_t)steps, S = (size_t)samples;
    size_t t = 0;
    auto enc = [&](size_t N) { const size_t C = std::min<size_t>((size_t)enc_chunk, N); t += al(C * 576 * 4) + 2 * al(C * 256 * 4); };
    t += al(R * 32 * 4) + al(R * 16 * 4);                     // enc0, x
    enc(R);                                                   // root encode
    t += al(D * 2 * S * R * 32 * 4) + al(D * 3 * S * R * 16 * 4) + al(2 * R * 16 * 4) + al(D * 3 * S * R * 4)
       + al(D * S * R * 4096 * 4) + al(D * S * R * 32 * 4) + al(3 * R * 4);
    t += D * 2 * al(2 * S * R * 512 * 4);                     // h1, h2 per stage
    {   const size_t N = D * 3 * S * R, C = std::min<size_t>((size_t)dec_chunk, N);
        t += 2 * al(N * 256 * 4) + al(C * 16384 * 4) + al(C * 65536 * 4) + al(((N + C - 1) / C) * 4); }
    enc(D * S * R);
    return (int64_t)(t + ((size_t)1 << 20));
}

int efe_arena_stats(efe_ctx* ctx, int64_t* capacity_bytes, int64_t* high_water_bytes, int64_t* grow_count) {
    EFE_LOCK(ctx);
    size_t have = 0;
    for (auto& b : ctx->arena.blocks) have += b.second;
    if (capacity_bytes) *capacity_bytes = (int64_t)have;
    if (high_water_bytes) *high_water_bytes = (int64_t)ctx->high_water;
    if (grow_count) *grow_count = ctx->arena_grows;
    return 0;
}

}  // extern "C"
