// torch.ops.efe.* : PyTorch-ROCm custom-op registration of the EFE engine, a thin layer over the C ABI of
// include/efe_engine.h.  Plain C++ (no device code): tensors in, tensors out; device memory comes from torch's allocator, the
// launch stream is torch's current HIP stream, errors become c10::Error (RuntimeError in Python).  The engine context is passed as
// an integer handle (the efe_ctx* returned by efe_create); its packed weights live in the context, not in the schema.
//
// Each op replaces one method of the reference's ActiveInferenceModel (/root/reference/src/torchmodel.py), cited per op.
// Only the CUDA (= HIP on ROCm) dispatch key is registered: calling an op with CPU tensors raises NotImplementedError -- there is
// no CPU fallback.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>

#include "efe_engine.h"

#ifndef EFE_OPS_BUILD_ID
#define EFE_OPS_BUILD_ID "unstamped"
#endif
extern "C" const char* efe_ops_build_id(void) {
    static const char stamp[] = "EFE_OPS_BUILD_ID=" EFE_OPS_BUILD_ID;
    return stamp + 17;
}

namespace {

using at::Tensor;
using OptT = c10::optional<Tensor>;

// The integer handle is checked against the engine's registry of live contexts (efe_ctx_alive: efe_create adds, efe_destroy removes), so a
// stale or made-up handle is a RuntimeError, not a dereference of freed memory.  CTX() also notes the context's device: every tensor of
// the call must live on it (in() below) -- the launch stream is torch's current stream of THAT device.
thread_local int tl_ctx_device = -1;
efe_ctx* CTX(int64_t h) {
    TORCH_CHECK(h != 0, "efe: null engine context");
    efe_ctx* c = reinterpret_cast<efe_ctx*>(static_cast<intptr_t>(h));
    TORCH_CHECK(efe_ctx_alive(c), "efe: stale or invalid engine context handle ", h, " (the context was destroyed, or this is not a handle of efe_create)");
    int dev = -1;
    TORCH_CHECK(efe_get_device(c, &dev, nullptr, 0) == 0, "efe: efe_get_device failed");
    tl_ctx_device = dev;
    return c;
}
void ok(efe_ctx* c, int rc) { TORCH_CHECK(rc == 0, "efe engine: ", efe_last_error(c)); }

Tensor in(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), "efe: ", name, " must be a HIP device tensor (there is no CPU fallback)");
    TORCH_CHECK((int)t.device().index() == tl_ctx_device, "efe: ", name, " is on device ", (int)t.device().index(), ", the engine context lives on device ",
                tl_ctx_device, " (one context per device: create the model on the tensor's device)");
    TORCH_CHECK(t.scalar_type() == at::kFloat, "efe: ", name, " must be float32");
    return t.contiguous();
}
const float* optp(const OptT& t, Tensor& keep, const char* name, int64_t numel) {
    if (!t.has_value() || !t->defined()) return nullptr;
    keep = in(*t, name);
    TORCH_CHECK(keep.numel() == numel, "efe: ", name, " has ", keep.numel(), " elements, expected ", numel);
    return keep.data_ptr<float>();
}
// the optional row set of a call (efe_rows): mask = uint8 [entries of the un-compacted batch], ids = int32 [entries of this call]
struct RowsArg { efe_rows r{nullptr, nullptr, 1, 0}; Tensor mk, ik; const efe_rows* ptr = nullptr; };
void rows_arg(RowsArg& ra, const OptT& mask, const OptT& ids, int64_t rows_per_entry, int64_t n_entries, int64_t n_total) {
    const bool hm = mask.has_value() && mask->defined(), hi = ids.has_value() && ids->defined();
    if (!hm && !hi) return;
    TORCH_CHECK(rows_per_entry >= 1, "efe: rows_per_entry must be >= 1");
    ra.r.rows_per_entry = (int32_t)rows_per_entry;
    if (hm) {
        TORCH_CHECK(mask->is_cuda() && mask->scalar_type() == at::kByte && mask->is_contiguous(), "efe: row mask must be a contiguous uint8 HIP tensor");
        TORCH_CHECK((int)mask->device().index() == tl_ctx_device, "efe: row mask on device ", (int)mask->device().index(), ", engine context on device ", tl_ctx_device);
        TORCH_CHECK(hi || mask->numel() >= n_entries, "efe: row mask has ", mask->numel(), " entries, the call has ", n_entries);
        // the mask is indexed by entry ID: it must cover the whole un-compacted batch (n_total; with ids and no n_total the bound is unknown here)
        TORCH_CHECK(n_total <= 0 || mask->numel() >= n_total, "efe: row mask has ", mask->numel(), " entries, the un-compacted batch has ", n_total);
        if (n_total <= 0 && !hi) n_total = mask->numel();
        ra.mk = *mask; ra.r.mask = ra.mk.data_ptr<uint8_t>();
    }
    if (hi) {
        TORCH_CHECK(ids->is_cuda() && ids->scalar_type() == at::kInt && ids->is_contiguous(), "efe: row ids must be a contiguous int32 HIP tensor");
        TORCH_CHECK((int)ids->device().index() == tl_ctx_device, "efe: row ids on device ", (int)ids->device().index(), ", engine context on device ", tl_ctx_device);
        TORCH_CHECK(ids->numel() == n_entries, "efe: row ids has ", ids->numel(), " entries, the call has ", n_entries);
        ra.ik = *ids; ra.r.ids = ra.ik.data_ptr<int32_t>();
    }
    TORCH_CHECK(n_total >= 0 && n_total <= INT32_MAX, "efe: bad n_total");
    ra.r.n_total = (int32_t)n_total;
    ra.ptr = &ra.r;
}
void* stream_of(const Tensor& t) {
    TORCH_CHECK((int)t.device().index() == tl_ctx_device, "efe: tensor on device ", (int)t.device().index(), ", engine context on device ", tl_ctx_device);
    return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}
efe_noise noise(int64_t seed, int64_t stage, int64_t pass, int64_t sample, int64_t row_offset) {
    efe_noise nz;
    nz.seed = (uint64_t)seed; nz.stage = (uint32_t)stage; nz.pass = (uint32_t)pass; nz.sample = (uint32_t)sample;
    nz.row_offset = (uint32_t)row_offset;
    return nz;
}
struct Geo { int s = 10, A = 4, C = 1, R = 64; int64_t img() const { return (int64_t)C * R * R; } };
Geo geo(efe_ctx* c) { Geo g; efe_get_config(c, &g.s, &g.A, &g.C, &g.R); return g; }
int rows(const Tensor& t, int64_t width, const char* name) {
    TORCH_CHECK(t.numel() % width == 0 && t.numel() > 0, "efe: ", name, " must be [M, ", width, "]");
    return (int)(t.numel() / width);
}

// ModelMid.transition_with_sample, torchmodel.py:58-66
std::tuple<Tensor, Tensor, Tensor> transition(int64_t h, const Tensor& pi_, const Tensor& s0_, int64_t seed, int64_t stage, int64_t pass,
                                              int64_t sample, int64_t row_offset, const OptT& eps) {
    efe_ctx* c = CTX(h);
    Tensor pi = in(pi_, "pi"), s0 = in(s0_, "s0"), ek;
    const Geo gq = geo(c);
    const int M = rows(s0, 10, "s0");
    TORCH_CHECK(pi.numel() == (int64_t)M * gq.A, "efe: pi must be [M, pi_dim]");
    Tensor ps1 = at::empty({M, 10}, s0.options()), mean = at::empty({M, 10}, s0.options()), lv = at::empty({M, 10}, s0.options());
    efe_noise nz = noise(seed, stage, pass, sample, row_offset);
    ok(c, efe_transition(c, pi.data_ptr<float>(), s0.data_ptr<float>(), M, &nz, optp(eps, ek, "eps", (int64_t)M * 10), ps1.data_ptr<float>(),
                         mean.data_ptr<float>(), lv.data_ptr<float>(), stream_of(s0)));
    return {ps1, mean, lv};
}

// ModelDown.decoder, torchmodel.py:139-141
Tensor decoder(int64_t h, const Tensor& s_, int64_t seed, int64_t stage, int64_t pass, int64_t sample, int64_t row_offset) {
    efe_ctx* c = CTX(h);
    Tensor s = in(s_, "s");
    const Geo gq = geo(c);
    const int M = rows(s, 10, "s");
    Tensor po = at::empty({M, gq.C, gq.R, gq.R}, s.options());
    efe_noise nz = noise(seed, stage, pass, sample, row_offset);
    ok(c, efe_decoder(c, s.data_ptr<float>(), M, &nz, po.data_ptr<float>(), stream_of(s)));
    return po;
}

// ModelDown.encoder / encoder_with_sample, torchmodel.py:134-137, 143-146
std::tuple<Tensor, Tensor, Tensor> encoder(int64_t h, const Tensor& o_, int64_t seed, int64_t stage, int64_t pass, int64_t sample,
                                           int64_t row_offset, const OptT& eps, bool want_s) {
    efe_ctx* c = CTX(h);
    Tensor o = in(o_, "o"), ek;
    const int M = rows(o, geo(c).img(), "o");
    Tensor s = want_s ? at::empty({M, 10}, o.options()) : at::empty({0}, o.options());
    Tensor mean = at::empty({M, 10}, o.options()), lv = at::empty({M, 10}, o.options());
    efe_noise nz = noise(seed, stage, pass, sample, row_offset);
    ok(c, efe_encoder(c, o.data_ptr<float>(), M, &nz, optp(eps, ek, "eps", (int64_t)M * 10), want_s ? s.data_ptr<float>() : nullptr,
                      mean.data_ptr<float>(), lv.data_ptr<float>(), stream_of(o)));
    return {s, mean, lv};
}

// ModelTop.encode_s, torchmodel.py:27-31
std::tuple<Tensor, Tensor, Tensor> habit(int64_t h, const Tensor& s_) {
    efe_ctx* c = CTX(h);
    Tensor s = in(s_, "s");
    const int M = rows(s, 10, "s");
    const int A = geo(c).A;
    Tensor logits = at::empty({M, A}, s.options()), q = at::empty({M, A}, s.options()), logq = at::empty({M, A}, s.options());
    ok(c, efe_habit(c, s.data_ptr<float>(), M, logits.data_ptr<float>(), q.data_ptr<float>(), logq.data_ptr<float>(), stream_of(s)));
    return {logits, q, logq};
}

// calculate_G / calculate_G_mean, torchmodel.py:270-327
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> calculate_g(int64_t h, const Tensor& s0_, const Tensor& pi0_, int64_t samples,
                                                                       bool mean_mode, int64_t seed, int64_t stage, int64_t row_offset,
                                                                       const OptT& eps, const OptT& mask, const OptT& ids, int64_t rows_per_entry, int64_t n_total) {
    efe_ctx* c = CTX(h);
    Tensor s0 = in(s0_, "s0"), pi0 = in(pi0_, "pi0"), ek;
    const Geo gq = geo(c);
    const int M = rows(s0, 10, "s0");
    TORCH_CHECK(pi0.numel() == (int64_t)M * gq.A, "efe: pi0 must be [M, pi_dim]");
    TORCH_CHECK(samples >= 1 && samples <= 65535, "efe engine: samples must be in [1, 65535]");
    const int S = mean_mode ? 1 : (int)samples;
    auto op = s0.options();
    Tensor G = at::empty({M}, op), terms = at::empty({3, M}, op), ps1 = at::empty({M, 10}, op), ps1m = at::empty({M, 10}, op),
           po1 = at::empty({M, gq.C, gq.R, gq.R}, op), parts = at::empty({2, M}, op);
    efe_noise nz = noise(seed, stage, 0, 0, row_offset);
    RowsArg ra;
    TORCH_CHECK(rows_per_entry < 1 || M % rows_per_entry == 0, "efe: rows_per_entry must divide the row count");
    rows_arg(ra, mask, ids, rows_per_entry, rows_per_entry >= 1 ? M / rows_per_entry : M, n_total);
    ok(c, efe_calculate_g_rows(c, s0.data_ptr<float>(), pi0.data_ptr<float>(), M, S, mean_mode ? 1 : 0, &nz, optp(eps, ek, "eps", (int64_t)3 * S * M * 10),
                          ra.ptr, G.data_ptr<float>(), terms.data_ptr<float>(), ps1.data_ptr<float>(), ps1m.data_ptr<float>(), po1.data_ptr<float>(),
                          parts.data_ptr<float>(), stream_of(s0)));
    return {G, terms, ps1, ps1m, po1, parts};
}

// calculate_G_repeated / calculate_G_4_repeated, torchmodel.py:227-268 (one row = one EFE rollout)
std::tuple<Tensor, Tensor, Tensor> rollout(int64_t h, const Tensor& o_, const Tensor& pi_, int64_t steps, int64_t samples, bool calc_mean,
                                           bool per_stage_mean, int64_t seed, int64_t stage, int64_t row_offset, const OptT& eps) {
    efe_ctx* c = CTX(h);
    Tensor o = in(o_, "o"), pi = in(pi_, "pi"), ek;
    const Geo gq = geo(c);
    const int M = rows(o, gq.img(), "o");
    TORCH_CHECK(pi.numel() == (int64_t)M * gq.A, "efe: o and pi must have the same number of rows");
    TORCH_CHECK(steps >= 1 && samples >= 1 && samples <= 65535, "efe engine: steps and samples must be >= 1");
    const int64_t S = (per_stage_mean && calc_mean) ? 1 : samples;
    auto op = o.options();
    Tensor G = at::empty({M}, op), terms = at::empty({3, M}, op), po1 = at::empty({M, gq.C, gq.R, gq.R}, op);
    efe_noise nz = noise(seed, stage, 0, 0, row_offset);
    ok(c, efe_rollout(c, o.data_ptr<float>(), pi.data_ptr<float>(), M, (int)steps, (int)samples, calc_mean ? 1 : 0, per_stage_mean ? 1 : 0, &nz,
                      optp(eps, ek, "eps", (int64_t)M * 10 + steps * 3 * S * M * 10), G.data_ptr<float>(), terms.data_ptr<float>(),
                      po1.data_ptr<float>(), stream_of(o)));
    return {G, terms, po1};
}

// calculate_G_given_trajectory, torchmodel.py:329-352
Tensor trajectory(int64_t h, const Tensor& s0_, const Tensor& ps1_, const Tensor& mean_, const Tensor& lv_, const Tensor& pi0_, int64_t seed,
                  int64_t stage, int64_t row_offset, const OptT& eps) {
    efe_ctx* c = CTX(h);
    Tensor s0 = in(s0_, "s0_traj"), ps1 = in(ps1_, "ps1_traj"), mean = in(mean_, "ps1_mean_traj"), lv = in(lv_, "ps1_logvar_traj"),
           pi0 = in(pi0_, "pi0_traj"), ek;
    const int T = rows(s0, 10, "s0_traj");
    TORCH_CHECK(ps1.numel() == (int64_t)T * 10 && mean.numel() == (int64_t)T * 10 && lv.numel() == (int64_t)T * 10 && pi0.numel() == (int64_t)T * geo(c).A,
                "efe: trajectory tensors must all have T rows");
    Tensor G = at::empty({T}, s0.options());
    efe_noise nz = noise(seed, stage, 0, 0, row_offset);
    ok(c, efe_trajectory(c, s0.data_ptr<float>(), ps1.data_ptr<float>(), mean.data_ptr<float>(), lv.data_ptr<float>(), pi0.data_ptr<float>(), T,
                         &nz, optp(eps, ek, "eps", (int64_t)3 * T * 10), G.data_ptr<float>(), stream_of(s0)));
    return G;
}

// mcts_step_simulate for E lock-step episodes, torchmodel.py:354-393
std::tuple<Tensor, Tensor, Tensor> simulate(int64_t h, const Tensor& s_, int64_t depth, bool use_means, int64_t seed, int64_t stage,
                                            int64_t row_offset, const OptT& eps, const OptT& u, const OptT& mask, const OptT& ids, int64_t n_total) {
    efe_ctx* c = CTX(h);
    Tensor s = in(s_, "starting_s"), ek, uk;
    const int E = rows(s, 10, "starting_s");
    TORCH_CHECK(depth >= 1 && depth <= 65535, "efe engine: depth must be in [1, 65535]");
    auto op = s.options();
    const int A = geo(c).A;
    Tensor G = at::empty({E}, op), pi0 = at::empty({E, depth, A}, op), q0 = at::empty({E, A}, op);
    efe_noise nz = noise(seed, stage, 0, 0, row_offset);
    RowsArg ra;
    rows_arg(ra, mask, ids, 1, E, n_total);
    ok(c, efe_simulate_rows(c, s.data_ptr<float>(), E, (int)depth, use_means ? 1 : 0, &nz, optp(eps, ek, "eps", (int64_t)4 * depth * E * 10),
                       optp(u, uk, "u", depth * E), ra.ptr, G.data_ptr<float>(), pi0.data_ptr<float>(), q0.data_ptr<float>(), stream_of(s)));
    return {G, pi0, q0};
}

// softmax_multi_with_log(-sum_G, n), /root/reference/src/util.py:46-53,68
std::tuple<Tensor, Tensor> action_posterior(int64_t h, const Tensor& g_, int64_t n, double temperature) {
    efe_ctx* c = CTX(h);
    Tensor g = in(g_, "sum_G");
    TORCH_CHECK(n >= 1 && n <= 8 && g.numel() % n == 0 && g.numel() > 0, "efe: sum_G must hold groups of n <= 8 values");
    const int64_t groups = g.numel() / n;
    Tensor P = at::empty({groups, n}, g.options()), logP = at::empty({groups, n}, g.options());
    ok(c, efe_action_posterior(c, g.data_ptr<float>(), (int)groups, (int)n, (float)temperature, P.data_ptr<float>(), logP.data_ptr<float>(),
                               stream_of(g)));
    return {P, logP};
}

// ActiveInferenceModel.check_reward, torchmodel.py:210-212
Tensor check_reward(int64_t h, const Tensor& o_) {
    efe_ctx* c = CTX(h);
    Tensor o = in(o_, "o");
    const int M = rows(o, geo(c).img(), "o");
    Tensor out = at::empty({M}, o.options());
    ok(c, efe_check_reward(c, o.data_ptr<float>(), M, out.data_ptr<float>(), stream_of(o)));
    return out;
}

// Model{Mid,Down}.reparameterize, torchmodel.py:54-56 / 130-132
Tensor reparameterize(int64_t h, const Tensor& mean_, const Tensor& lv_, int64_t seed, int64_t stage, int64_t pass, int64_t sample,
                      int64_t row_offset, const OptT& eps) {
    efe_ctx* c = CTX(h);
    Tensor mean = in(mean_, "mean"), lv = in(lv_, "logvar"), ek;
    TORCH_CHECK(mean.dim() == 2 && mean.sizes() == lv.sizes(), "efe: mean and logvar must be [M, n]");
    const int M = (int)mean.size(0), n = (int)mean.size(1);
    Tensor out = at::empty({M, n}, mean.options());
    efe_noise nz = noise(seed, stage, pass, sample, row_offset);
    ok(c, efe_reparameterize(c, mean.data_ptr<float>(), lv.data_ptr<float>(), M, n, &nz, optp(eps, ek, "eps", (int64_t)M * n), out.data_ptr<float>(),
                             stream_of(mean)));
    return out;
}

}  // namespace

TORCH_LIBRARY(efe, m) {
    m.def("transition(int ctx, Tensor pi, Tensor s0, int seed, int stage, int pass_id, int sample, int row_offset, Tensor? eps) -> (Tensor ps1, Tensor mean, Tensor logvar)");
    m.def("decoder(int ctx, Tensor s, int seed, int stage, int pass_id, int sample, int row_offset) -> Tensor");
    m.def("encoder(int ctx, Tensor o, int seed, int stage, int pass_id, int sample, int row_offset, Tensor? eps, bool want_s) -> (Tensor s, Tensor mean, Tensor logvar)");
    m.def("habit(int ctx, Tensor s) -> (Tensor logits, Tensor q, Tensor logq)");
    m.def("calculate_g(int ctx, Tensor s0, Tensor pi0, int samples, bool mean_mode, int seed, int stage, int row_offset, Tensor? eps, Tensor? mask=None, Tensor? ids=None, int rows_per_entry=1, int n_total=0) -> (Tensor G, Tensor terms, Tensor ps1, Tensor ps1_mean, Tensor po1, Tensor t2parts)");
    m.def("rollout(int ctx, Tensor o, Tensor pi, int steps, int samples, bool calc_mean, bool per_stage_mean, int seed, int stage, int row_offset, Tensor? eps) -> (Tensor sum_G, Tensor sum_terms, Tensor po1)");
    m.def("trajectory(int ctx, Tensor s0_traj, Tensor ps1_traj, Tensor ps1_mean_traj, Tensor ps1_logvar_traj, Tensor pi0_traj, int seed, int stage, int row_offset, Tensor? eps) -> Tensor");
    m.def("simulate(int ctx, Tensor starting_s, int depth, bool use_means, int seed, int stage, int row_offset, Tensor? eps, Tensor? u, Tensor? mask=None, Tensor? ids=None, int n_total=0) -> (Tensor G, Tensor pi0, Tensor Qpi0)");
    m.def("action_posterior(int ctx, Tensor sum_G, int n, float temperature) -> (Tensor P, Tensor logP)");
    m.def("check_reward(int ctx, Tensor o) -> Tensor");
    m.def("reparameterize(int ctx, Tensor mean, Tensor logvar, int seed, int stage, int pass_id, int sample, int row_offset, Tensor? eps) -> Tensor");
}

TORCH_LIBRARY_IMPL(efe, CUDA, m) {       // the CUDA dispatch key is the HIP device on ROCm builds of PyTorch
    m.impl("transition", &transition);
    m.impl("decoder", &decoder);
    m.impl("encoder", &encoder);
    m.impl("habit", &habit);
    m.impl("calculate_g", &calculate_g);
    m.impl("rollout", &rollout);
    m.impl("trajectory", &trajectory);
    m.impl("simulate", &simulate);
    m.impl("action_posterior", &action_posterior);
    m.impl("check_reward", &check_reward);
    m.impl("reparameterize", &reparameterize);
}
