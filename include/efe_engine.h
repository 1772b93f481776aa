/* efe_engine.h -- C ABI of the MI355X expected-free-energy (EFE) rollout engine.
 *
 * Drop-in boundary for the Monte-Carlo EFE hot path of zfountas/deep-active-inference-mc.
 * The reference has no FFI: its boundary is the Python object `ActiveInferenceModel`
 * (/root/reference/src/torchmodel.py:149-393).  Each entry point below replaces one method of that
 * object (cited per function); the Python mirror in deep-active-inference-mc_amd/model.py binds
 * them with ctypes and keeps the reference's method names, argument meaning and return tuples.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch types.  All tensor pointers are DEVICE pointers to
 *     contiguous fp32 unless the name ends in _host.  Observations are NCHW [M,1,64,64].
 *   - no ownership transfer: the caller allocates every input/output buffer; the engine owns only
 *     its packed weights and a scratch arena.  Calls are stream-ordered on `stream` (a hipStream_t
 *     passed as void*; NULL = default stream) and return without synchronising, unless the arena must
 *     grow (first call at a new size; efe_reserve pre-sizes it so that steady-state calls never hipMalloc).
 *   - threading / streams: a context has ONE scratch arena.  Entry points take a per-context mutex, so
 *     several host threads may share a context (their calls serialise); a call issued on a different
 *     stream than the previous call first waits (hipStreamWaitEvent) for that call's last kernel, so
 *     switching streams is safe and costs one event wait.  For concurrent execution on several streams
 *     use one context per stream (weights are 21 MB).
 *   - return value: 0 on success, non-zero on error (efe_last_error() gives the message).
 *   - noise: MC-dropout masks / normals / action uniforms are a pure function of
 *     (seed, stage, pass, sample, global row = row_offset + r, element) -- see csrc/philox.h --
 *     so results are independent of batching and of the number of GPUs.  `stage` is the caller's
 *     call counter: one calculate_G call (or one stage of a rollout) consumes one stage value.
 *     `eps` pointers are optional injected normals (NULL = generated on device).
 */
#ifndef EFE_ENGINE_H
#define EFE_ENGINE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct efe_ctx efe_ctx;

/* lifecycle ------------------------------------------------------------------------------------ */
int efe_create(efe_ctx** out, int device);                       /* ActiveInferenceModel.__init__, torchmodel.py:150-165 (10, 4, 1, 64) */
/* ActiveInferenceModel(s_dim, pi_dim, ..., colour_channels, resolution), torchmodel.py:150: s_dim must be 10 (the reference never uses
 * another), pi_dim 2..6, channels 1..3, resolution a multiple of 4 in [32, 128].  (1, 64) is the Dynamic-dSprites geometry on the fused
 * kernels; any other geometry (BASELINE configs[4]: pi 3, 3 x 84 x 84) is BUILD-DEFINED -- the reference rejects it (torchmodel.py:77-82)
 * and its reward (calc_reward_animalai, torchmodel.py:214) is undefined -- and runs the generic convolution path: encoder
 * Conv2d(k3,s2) x 4 + dense head with 64 * h4 * h4 inputs, decoder dense head -> Linear(256, 64 * (res/4)^2) -> ConvT(64,64,s1) ->
 * ConvT(64,64,s2) -> ConvT(64,32,s2) -> ConvT(32,C,s1) + sigmoid (resolution 32 = the reference's own variant, torchmodel.py:77-80: decoder base
 * 16 x 16 and a stride-1 third layer; its NETWORKS are pinned against the reference, tests/golden/nets32_*.npz), reward = SUM over (c,h,w) of the NCHW-broadcast log-likelihood of
 * torchutils.py:34-37.  Observations are NCHW [M, C, res, res].  Parity unpinned: validated against oracle/efe_oracle.py (cfg=) only. */
int efe_create_cfg(efe_ctx** out, int device, int s_dim, int pi_dim, int channels, int resolution);
int efe_get_config(efe_ctx* ctx, int* s_dim, int* pi_dim, int* channels, int* resolution);      /* outputs may be NULL */
/* the HIP device the context was created on (efe_create's `device`) and the PCI bus id string of that device ("0000:c1:00.0"; buf may be
 * NULL): what a multi-GPU launcher checks so that rank r really owns GPU r (bench.py gathers them and refuses two ranks on one device) */
int efe_get_device(efe_ctx* ctx, int* device, char* pci_bus_id, int pci_bus_id_len);
void efe_destroy(efe_ctx* ctx);                                  /* a handle that is not live (already destroyed, never created) is ignored */
const char* efe_last_error(efe_ctx* ctx);
/* 1 if `ctx` is a live context of this process (created by efe_create[_cfg], not yet destroyed), else 0; never dereferences the pointer.
 * The library keeps a registry of its contexts and EVERY entry point checks its handle against it first (a stale handle is return code 1,
 * not a use of freed memory); bindings that carry the handle as an integer (torch.ops.efe.*, ctypes) use this to raise a proper error.
 * Every entry point also restores the CALLER's current HIP device before it returns (the context's device is current only inside the call). */
int efe_ctx_alive(const efe_ctx* ctx);
int efe_abi_version(void);                                       /* 6 (history of the versions: INTEGRATION.md section 4) */
/* hex digest of the sources this library was compiled from (build.py stamps it; the Python loader refuses a library whose
 * digest differs from the sources next to it, so a stale shipped binary fails loudly). */
const char* efe_build_id(void);

/* weights: reference state_dict tensors (host, reference layout), key = "<top|mid|down>.<state_dict key>",
 * e.g. "down.po_net.13.weight" (ConvTranspose2d [Cin,Cout,3,3]).  Replaces load_weights,
 * torchmodel.py:173-177 (the .pth unpickling stays in Python).  efe_commit_weights packs them into the
 * MFMA fragment-major device layout.  The host copies are kept: after a commit a caller may update any subset of
 * tensors with efe_set_weight and commit again; the packed buffers of the previous commit are freed (no growth). */
int efe_set_weight(efe_ctx* ctx, const char* key, const float* data_host, const int64_t* shape, int ndim);
int efe_commit_weights(efe_ctx* ctx);

/* options
 *   launch groups : "dec_chunk" (decoder rows per launch group), "enc_chunk", "dec_chunk_g" / "dec_budget_g" (generic-geometry decoder: cap in
 *                   images / in bytes of layer activations per launch group; 0.68 MB per image at 84 x 84, default budget 28 GiB)
 *   semantics     : "reward_upstream_intent" (0 / 1, default 0).  0 = the reward the shipped port computes (torchutils.py:34-37 on NCHW input: target
 *                   1 for image rows h < H/2, every pixel counts; pinned by the oracle).  1 = what the upstream NHWC code means:
 *                   only the top three rows count, target 1 on their left half; dSprites: mean over those 192 pixels * 10, other geometries: sum.
 *   A / B         : "fuse_final_g" (generic path: last two decoder layers in one kernel, default 1), "sim_split" (efe_simulate of <= 16 episodes: the habit-policy chain on eight workgroups per 8 episodes that split
 *                   the two wide transition layers, default 1, bit-identical to 0), "enc_tiled" (generic path, encoder layers 1 and 2: 2 = one
 *                   kernel with conv1 kept in LDS (default), 1 = LDS-tiled, one launch per layer, 0 = the direct kernel for every layer; bit-identical),
 *                   "ct_fuse12" (generic path: the decoder's first two ConvTranspose layers in one kernel, layer 1's output kept in LDS, default 1,
 *                   bit-identical to 0), "mid_unfused" (layer-by-layer transition MLP), "head_unfused" (the decoder / encoder dense heads as
 *                   one k_dense launch per layer instead of one k_head launch per head; same masks, fp32 summation order differs).  None of them removes work:
 *                   every setting computes the same quantities (fp32 summation order may differ where stated).
 *   experiment    : "mfma_bf16x3" (0 / 1, default 0; Dynamic-dSprites geometry only).  1 = the decoder's Linear(256, 16384) and its first two
 *                   ConvTranspose layers run on the bf16 matrix pipe with both operands split into three bf16 planes (six products per fp32
 *                   product, fp32 accumulation: csrc/bf16x3.hip).  Inputs narrower than the reference's fp32 arithmetic, results inside the same
 *                   tolerances (every fixture is run through it); never the default, never the benchmark's headline.  With it, results are no
 *                   longer bit-identical across launch sizes (launches of <= 128 images keep the fp32 small-launch kernels), only within tolerance.
 *   development   : "poison" (pre-fill scratch with a byte), "trace" (synchronise and log every profiled launch), "arena_align", "check_rows"
 *                   (range-check efe_rows.ids against efe_rows.n_total on the host before every _rows call: one synchronisation per call) */
int efe_set_option(efe_ctx* ctx, const char* name, int64_t value);

/* scratch arena: efe_reserve makes the arena one block of >= bytes (synchronises once); efe_rollout_scratch_bytes is what
 * efe_rollout(M, steps, samples) needs with the current chunk options; efe_arena_stats reports capacity, the largest use
 * of any call so far and how many hipMalloc calls the arena has made (outputs may be NULL). */
int efe_reserve(efe_ctx* ctx, int64_t bytes);
int64_t efe_rollout_scratch_bytes(efe_ctx* ctx, int M, int steps, int samples);
int efe_arena_stats(efe_ctx* ctx, int64_t* capacity_bytes, int64_t* high_water_bytes, int64_t* grow_count);

/* The row set of ONE efe_calculate_g_rows / efe_simulate_rows call (ABI 4) -- the lock-step planner's early-stopped episodes (the
 * per-episode break of /root/reference/src/mcts.py:176).  The rows of a call are grouped into ENTRIES of rows_per_entry consecutive rows
 * (efe_calculate_g_rows: the pi_dim action rows of an episode; efe_simulate_rows: one episode = one entry, rows_per_entry ignored).
 *   mask : DEVICE array, one byte per entry ID, read when the kernels run (earlier work on the same stream may update it); the per-image
 *          kernels (decoder stages, encoder trunk: ~90 % of the work) skip dead entries, whose outputs are then unspecified; live rows
 *          are bit-identical to an unmasked call.  NULL = all live.
 *   ids  : DEVICE array, entry slot -> entry ID: the call's entry i IS entry ids[i] of the un-compacted batch -- its noise keys are those of
 *          rows ids[i] * rows_per_entry + k (plus efe_noise.row_offset) and the mask is read at ids[i].  A planner that has lost episodes
 *          passes only the live ones (a dense, smaller call) and still draws exactly what the full batch would.  NULL = identity.
 *   n_total : entries of the UN-COMPACTED batch = the length of `mask` and the exclusive upper bound of every id (ABI 5).  0 = not stated.
 *          When stated, a call with more entries than n_total (no ids) fails, and with the development option "check_rows" = 1 the
 *          ids are copied back and range-checked before the launch (one synchronisation: a stale or corrupt id would otherwise be a
 *          silent out-of-bounds read of `mask` and a wrong noise key).
 * A NULL efe_rows* means "all rows, identity" (the context-state shim efe_set_row_mask of ABI 2 - 5 is gone in ABI 6: the row set is an argument). */
typedef struct efe_rows {
    const uint8_t* mask;
    const int32_t* ids;
    int32_t rows_per_entry;
    int32_t n_total;
} efe_rows;

typedef struct efe_noise {
    uint64_t seed;
    uint32_t stage;       /* call / stage counter */
    uint32_t pass;        /* network-level calls only: which pass id keys the masks (csrc/philox.h) */
    uint32_t sample;      /* network-level calls only */
    uint32_t row_offset;  /* global index of local row 0 */
} efe_noise;

/* network level ---------------------------------------------------------------------------------- */
/* ModelMid.transition_with_sample, torchmodel.py:58-66.  eps: optional [M,10]. */
int efe_transition(efe_ctx*, const float* pi /*[M,4]*/, const float* s0 /*[M,10]*/, int M, const efe_noise* nz,
                   const float* eps, float* ps1, float* mean, float* logvar, void* stream);
/* ModelDown.decoder, torchmodel.py:139-141.  po: [M,1,64,64]. */
int efe_decoder(efe_ctx*, const float* s /*[M,10]*/, int M, const efe_noise* nz, float* po, void* stream);
/* ModelDown.encoder / encoder_with_sample, torchmodel.py:134-137,143-146.  s may be NULL. */
int efe_encoder(efe_ctx*, const float* o /*[M,1,64,64]*/, int M, const efe_noise* nz, const float* eps,
                float* s, float* mean, float* logvar, void* stream);
/* ModelTop.encode_s, torchmodel.py:27-31 (no dropout). */
int efe_habit(efe_ctx*, const float* s /*[M,10]*/, int M, float* logits, float* q, float* logq, void* stream);

/* ActiveInferenceModel.check_reward, torchmodel.py:210-212 (resolution-64 branch). o: [M,1,64,64] -> out [M]. */
int efe_check_reward(efe_ctx*, const float* o, int M, float* out, void* stream);
/* Model{Mid,Down}.reparameterize, torchmodel.py:54-56 / 130-132: out = eps * exp(logvar/2) + mean, [M,n];
 * eps optional (NULL = Philox normals keyed by nz->pass / sample / stage / row_offset). */
int efe_reparameterize(efe_ctx*, const float* mean, const float* logvar, int M, int n, const efe_noise* nz, const float* eps,
                       float* out, void* stream);

/* EFE level -------------------------------------------------------------------------------------- */
/* calculate_G (torchmodel.py:270-300) when mean_mode == 0; calculate_G_mean (torchmodel.py:302-327)
 * when mean_mode == 1 (samples forced to 1).
 * eps: optional [3*samples, M, 10] = T1_0..T1_{S-1}, T2_0..T2_{S-1}, D2B_0..D2B_{S-1}.
 * outputs: G[M], terms[3,M], ps1[M,10] (last sample; NULL ok), ps1_mean[M,10], po1[M,1,64,64] (last sample; NULL ok),
 * t2parts[2,M] (term2_1, term2_2; NULL ok). */
int efe_calculate_g(efe_ctx*, const float* s0, const float* pi0, int M, int samples, int mean_mode,
                    const efe_noise* nz, const float* eps,
                    float* G, float* terms, float* ps1, float* ps1_mean, float* po1, float* t2parts, void* stream);
/* the same over the row set `rows` (Node.expand of the lock-step planner, /root/reference/src/mcts.py:64-86 for every live episode at
 * once): M rows = M / rows_per_entry entries; inputs, eps and outputs are in the call's (compact) row order.  rows == NULL: as above. */
int efe_calculate_g_rows(efe_ctx*, const float* s0, const float* pi0, int M, int samples, int mean_mode,
                         const efe_noise* nz, const float* eps, const efe_rows* rows,
                         float* G, float* terms, float* ps1, float* ps1_mean, float* po1, float* t2parts, void* stream);

/* calculate_G_repeated (torchmodel.py:227-245) when per_stage_mean == 0;
 * calculate_G_4_repeated (torchmodel.py:247-268) semantics when per_stage_mean == 1 (calc_mean then
 * switches every stage to calculate_G_mean).  One row = one "EFE rollout".
 * nz->stage = stage0; stage t uses stage0 + t; the root encode uses (stage0, PASS_ROOT).
 * eps: optional [M*10 (root)] followed by per stage [3*S, M, 10].
 * outputs: sum_G[M], sum_terms[3,M], po1[M,1,64,64] (NULL ok). */
int efe_rollout(efe_ctx*, const float* o, const float* pi, int M, int steps, int samples, int calc_mean,
                int per_stage_mean, const efe_noise* nz, const float* eps,
                float* sum_G, float* sum_terms, float* po1, void* stream);

/* calculate_G_given_trajectory (torchmodel.py:329-352): rows are trajectory steps. G[T]. */
int efe_trajectory(efe_ctx*, const float* s0_traj, const float* ps1_traj, const float* ps1_mean_traj,
                   const float* ps1_logvar_traj, const float* pi0_traj, int T, const efe_noise* nz, const float* eps,
                   float* G, void* stream);

/* mcts_step_simulate (torchmodel.py:354-393) for E lock-step episodes: habit-policy rollout of `depth`
 * steps from starting_s[E,10], then G over each trajectory.  Noise rows: steps use global row
 * row_offset+e (sample = t); trajectory rows use (row_offset+e)*depth + t.
 * eps: optional injected normals [depth][E][10] (the transition of step t) followed by [3][E*depth][10] (the trajectory's
 *      T1 slot (unused), T2, D2B -- the layout of efe_trajectory); u: optional injected action uniforms [depth][E].
 * outputs: G_mean[E], pi0[E,depth,4] one-hot, Qpi0[E,4] (habit posterior of the first step). */
int efe_simulate(efe_ctx*, const float* starting_s, int E, int depth, int use_means, const efe_noise* nz,
                 const float* eps, const float* u, float* G_mean, float* pi0, float* Qpi0, void* stream);
/* the same over the row set `rows` (entry = episode): episode slot e draws the noise of episode ids[e]; rows == NULL: as above. */
int efe_simulate_rows(efe_ctx*, const float* starting_s, int E, int depth, int use_means, const efe_noise* nz,
                      const float* eps, const float* u, const efe_rows* rows, float* G_mean, float* pi0, float* Qpi0, void* stream);

/* softmax_multi_with_log(-sum_G, n) (/root/reference/src/util.py:46-53,68): action posterior. */
int efe_action_posterior(efe_ctx*, const float* sum_G /*[n_groups*n]*/, int n_groups, int n, float temperature,
                         float* P, float* logP, void* stream);

/* Dynamic-dSprites environment, batched over E games (/root/reference/src/game_environment.py).
 * state [E,7] = (colour, shape, scale, orientation, x, y, accumulated reward), last_r [E]; all device pointers.
 * No engine weights are involved: these calls work on any context.
 *   efe_env_reset : randomize_environment_all (:72-75), latents / reward / last_r drawn from Philox(seed, stage).
 *   efe_env_new_image: new_image_all (:83-88): fresh latents from Philox(seed, stage); reward slot and last_r untouched
 *                   (what the reference constructor calls, :21).
 *   efe_env_step  : pi_to_action(actions[e], e, repeats) for every game (:113-169); a finished round resamples the
 *                   latents (new_image, :84-87) from Philox(seed, stage); round_changed [E] may be NULL.
 *   efe_env_render: s_to_o (:44-54): frames[e] = imgs[index(state[e])] (uint8 -> float, imgs is [n_imgs,64,64] uint8)
 *                   with the reward bar; err[e] = 1 where |last_r| > 1 (the reference raises ValueError), may be NULL. */
int efe_env_reset(efe_ctx*, float* state, float* last_r, int E, const efe_noise* nz, void* stream);
int efe_env_new_image(efe_ctx*, float* state, int E, const efe_noise* nz, void* stream);
int efe_env_step(efe_ctx*, float* state, float* last_r, const int32_t* actions, int E, int repeats, const efe_noise* nz,
                 int32_t* round_changed, void* stream);
int efe_env_render(efe_ctx*, const float* state, const float* last_r, const uint8_t* imgs, int64_t n_imgs, float* frames,
                   int32_t* err, int E, void* stream);

/* ---- device-resident tree of the lock-step MCTS planner (counterpart of Node / active_inference_mcts, src/mcts.py:11-195) ----
 * All pointers are device pointers owned by the caller.  tree = { W, N, Qpi [E][cap][A] floats, child [E][cap][A] int32
 * (-1 = unexpanded), S [E][cap][s_dim] }.  One thread per episode; every formula in the reference's fp32 order with torch's
 * NaN rules (an unvisited edge makes Q = 0/0).  No engine weights are involved.
 *   efe_mcts_select  : tree policy (Node.select / probs_for_selection, :36-57) for every active episode: path_nodes / path_act
 *                      [E][max_depth], path_len [E], leaf [E], the leaf's state [E][s_dim] and its A-fold repeat [E*A][s_dim]
 *   efe_mcts_expand  : Node.expand bookkeeping (:64-86) where mask[e]: W[leaf] -= G, N[leaf] += 1, A children with states ps_next
 *   efe_mcts_backprop: Qpi[leaf] = q0, g = mean_r sims[r][e], W -= g and N += 1 along the path (:91-99, 186-191);
 *                      g_out [E] and active_out [E] are the iteration's history row
 *   efe_mcts_stop    : early stop (:130-131, 176) active[e] &= !(max(N0/sum) - mean(N0/sum) > threshold), stop_at[e] = repeat
 *                      for the episodes that stop now, *n_active = episodes still active */
typedef struct efe_mcts_tree { float* W; float* N; float* Qpi; int32_t* child; float* S; int32_t E, cap, A, s_dim; } efe_mcts_tree;
/* efe_mcts_step (ABI 4): the tree work between two iterations' engine calls as ONE launch, per episode in this order: efe_mcts_expand of the
 * PREVIOUS iteration's leaf (ABI 6: prev_n_nodes / prev_G / prev_ps_next, all three or none -- NULL: the caller has run efe_mcts_expand itself),
 * efe_mcts_backprop of the PREVIOUS iteration (prev_path_len == NULL: none; path_nodes / leaf must still hold that iteration's selection),
 * efe_mcts_stop, efe_mcts_select.  n_active: a zero-initialised word of its own per iteration (no memset is issued). */
int efe_mcts_step(efe_ctx*, const efe_mcts_tree* tree, const int32_t* prev_path_act, const int32_t* prev_path_len, const float* sims, int n_sims,
                  const float* q0, float* prev_g_out, uint8_t* prev_active_out, uint8_t* active, int32_t* stop_at, int repeat, float threshold,
                  int32_t* n_active, float C, int use_prior, int max_depth, int32_t* path_nodes, int32_t* path_act, int32_t* path_len, int32_t* leaf,
                  float* leaf_s, float* leaf_s_rep, int32_t* prev_n_nodes, const float* prev_G, const float* prev_ps_next, void* stream);
int efe_mcts_select(efe_ctx*, const efe_mcts_tree* tree, const uint8_t* active, float C, int use_prior, int max_depth,
                    int32_t* path_nodes, int32_t* path_act, int32_t* path_len, int32_t* leaf, float* leaf_s, float* leaf_s_rep,
                    void* stream);
int efe_mcts_expand(efe_ctx*, const efe_mcts_tree* tree, int32_t* n_nodes, const int32_t* nodes, const uint8_t* mask, const float* G,
                    const float* ps_next, void* stream);
int efe_mcts_backprop(efe_ctx*, const efe_mcts_tree* tree, const int32_t* path_nodes, const int32_t* path_act, const int32_t* path_len,
                      const int32_t* leaf, const uint8_t* active, const float* sims, int n_sims, const float* q0, int max_depth,
                      float* g_out, uint8_t* active_out, void* stream);
int efe_mcts_stop(efe_ctx*, const efe_mcts_tree* tree, uint8_t* active, int32_t* stop_at, int repeat, float threshold,
                  int32_t* n_active, void* stream);


/* introspection for benches: algorithmic MACs of the last EFE-level call. */
int64_t efe_last_call_macs(efe_ctx*);

/* per-kernel-class timing with HIP events recorded on the launch stream (bench.py roofline leg).
 * classes: 0 transition MLP, 1 decoder dense 10-256-256-256, 2 decoder dense 256->16384, 3 unused,
 * 4 k_dec_a (ConvT 64->64 s1 + ConvT 64->64 s2), 5 k_dec_b (ConvT 64->32 s2 + final conv + sigmoid + reductions),
 * 6 unused, 7 encoder, 8 other.
 * efe_prof_read synchronises the device, returns summed milliseconds and launch counts per class, and clears. */
int efe_prof_enable(efe_ctx*, int on);   /* 0 = off, < 0 = every class, otherwise a bitmask (bit c = class c) */
int efe_prof_classes(void);
int efe_prof_read(efe_ctx*, double* ms /*[classes]*/, int64_t* launches /*[classes]*/);

#ifdef __cplusplus
}
#endif
#endif
