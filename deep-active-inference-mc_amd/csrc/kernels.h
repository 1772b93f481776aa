// Device kernels of the EFE rollout engine (gfx950 / CDNA4 only): shared argument structs and launchers.
//
// Every dense contraction on the hot path (reference layers: /root/reference/src/torchmodel.py:41-52 ps_net,
// :84-104 qs_net, :106-128 po_net, :19-25 qpi_net) has the same MFMA mapping:
//
//   * MFMA rows  = output features (A operand = weights, pre-packed fragment-major so every wave-level load is one
//                  fully coalesced 1 KiB global_load_dwordx4),
//   * MFMA cols  = batch rows / output pixels (B operand = NHWC activations, 16 B per lane, from LDS in the fused
//                  kernels or straight from L2 in k_dense),
//   * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD,
//   * a lane owns ONE batch row / pixel and 16 features per 32x32 tile, so the MC-dropout mask of a row is one Philox
//     call per 128 features, generated in the epilogue; no mask is ever stored.
//
// kernels.hip: k_dense + the small VALU kernels (reparameterisation, term combine, softmax, sampling);
// decoder.hip: k_fc4, k_dec_a, k_dec_b;  encoder.hip: k_enc_trunk;  mfma_pipe.h: the shared software pipeline.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "philox.h"

namespace efe {

// The batch of one launch is [group][row]: a group is one network evaluation ("pass") over the same
// rows_per_group logical rows.  Group g of a multi-stage batch decomposes as t = g / per_stage (stage),
// q = g % per_stage, pass = pass[q / S], sample = sample0 + q % S.
struct GroupMap {
    int per_stage, S;
    uint32_t pass[3];
    uint32_t stage0, sample0;
    // optional row identities (efe_rows.ids: the lock-step planner's compacted batches): logical row r of a group is row
    // ids[r / ids_div] * ids_div + r % ids_div of the un-compacted batch -- the noise keys follow the episode, not its slot
    const int32_t* ids = nullptr; int ids_div = 1;
};
// global noise row of logical row r of a group (counter word 1 of every Philox draw of that row)
__device__ __forceinline__ uint32_t global_row(const int32_t* ids, int ids_div, int r, uint32_t row_offset) {
    if (!ids) return row_offset + (uint32_t)r;
    const int e = r / ids_div;
    return row_offset + (uint32_t)(ids[e] * ids_div + (r - e * ids_div));
}
__host__ __device__ inline void group_decode(const GroupMap& gm, int g, int& t, int& pidx, int& samp) {
    t = g / gm.per_stage;
    const int q = g - t * gm.per_stage;
    pidx = q / gm.S;
    samp = q - pidx * gm.S;
}
__device__ __forceinline__ uint2 group_key(const GroupMap& gm, int g) {
    int t, pidx, samp;
    group_decode(gm, g, t, pidx, samp);
    const uint32_t pass = pidx == 0 ? gm.pass[0] : pidx == 1 ? gm.pass[1] : gm.pass[2];
    return make_uint2(stream_id(pass, gm.sample0 + (uint32_t)samp), gm.stage0 + (uint32_t)t);
}

// Optional liveness mask of the logical rows of a call (efe_rows.mask: the lock-step planner's early-stopped episodes): image m of
// a launch belongs to logical row (m0 + m) % rows_per_group and is evaluated iff mask == nullptr or mask[row / div] != 0.  The
// per-image kernels (decoder stages, encoder trunk) skip dead images; the outputs of dead rows are unspecified.
struct RowMask {
    const uint8_t* mask; int div, m0, rows_per_group;
    const int32_t* ids = nullptr;       // optional: entry slot -> entry id (the mask is indexed by id)
};
__device__ __forceinline__ bool row_live(const RowMask& k, int m) {
    if (!k.mask) return true;
    const int slot = ((k.m0 + m) % k.rows_per_group) / k.div;
    return k.mask[k.ids ? k.ids[slot] : slot] != 0;
}

// One pixel's term of the reward log-likelihood log_bernoulli(x = pr, p = target) (/root/reference/src/torchutils.py:30-37):
//   intent 0: what the shipped port computes -- on NCHW input the target broadcasts to 1 for image rows oh < H / 2 and 0 below, and
//             every pixel counts (pinned by the oracle);
//   intent 1: what the upstream NHWC code means -- only the top three rows (the reward bar) count, with target 1
//             on their left half and 0 on the right half.
__device__ __forceinline__ float reward_term(float pr, int oh, int ow, int H, int W, int intent) {
#pragma clang fp contract(off)
    const float D1 = 1.00001f, D0 = 0.00001f;
    const bool one = intent ? (ow < W / 2) : (oh < H / 2);
    // (explicit contraction: the same bits from every inlined copy)
    const float t = one ? __builtin_fmaf(pr, logf(D1), (1.0f - pr) * logf(D1 - 1.0f)) : __builtin_fmaf(pr, logf(D0), (1.0f - pr) * logf(D1));
    return (intent && oh >= 3) ? 0.0f : t;
}

struct GemmArgs {
    const float* Wp;      // packed weights [tap][mtile][kc][64 lanes][4]
    const float* bias;    // [mtiles*32]
    const float* X;       // input activations
    float* Y;             // output activations
    const float* zeros;   // >= 4 KiB of zeros (source for padded taps / out-of-range rows)
    int n_pix;            // GEMM columns in this launch (batch rows)
    int cin;              // K per tap (multiple of 8)
    int cout;             // real output features (multiple of 4)
    int mtiles;           // 32-feature tiles in the packed weights
    int ldx, ldy;         // row strides in floats
    int x_mod;            // if > 0 the input row is (m % x_mod)  (same input for every MC pass)
    int relu;
    // MC-dropout: keep-mask * 2.0 from Philox, keyed by (tag, global row, group key)
    int dropout;
    uint32_t tag, k0, k1;
    GroupMap gm;          // group index -> (stream id, stage)
    int rows_per_group;
    uint32_t row_offset;  // global index of row 0 of a group on this rank
    int m0;               // index of column 0 of this launch inside the [group][row] batch
    const void* Wb3 = nullptr;   // options mfma_bf16x3 / mfma_f16x2 (bf16x3.hip): the same weights as 16-bit planes [mtile][16][planes][64 lanes][8]
    int split = 0;               // 1 = three bf16 planes, 2 = two fp16 planes
    float wb3_scale_inv = 1.0f;  // fp16 split: 1 / (the power of two the packed weights were scaled by)
};

// fused decoder, stage A: x4 [rows][16][16][64] -> ConvT(64,64,s1)+ReLU -> ConvT(64,64,s2)+ReLU -> y2 [rows][4 parities][8 channel groups][16x16 positions][8] (see k_dec_a)
struct DecAArgs {
    const float* x4; float* y2;
    const float* w1; const float* b1;   // packed [9][2][8][64][4], bias [64]
    const float* w2; const float* b2;
    int rows;
    RowMask live;
    int parts;            // 1 = persistent, one image per workgroup pass; 8 = small launches, an image over eight workgroups (k_dec_a_s)
    int* queue;           // zero-initialised ticket counter of this launch: images beyond the first two per workgroup are claimed dynamically
    const void* w1b3 = nullptr; const void* w2b3 = nullptr;      // options mfma_bf16x3 / mfma_f16x2: the two layers' weights as 16-bit planes [tap][2][4][planes][64 lanes][8] (bf16x3.hip)
    int split = 0;                                               // 1 = three bf16 planes, 2 = two fp16 planes
    float w1s = 1.0f, w1s_inv = 1.0f, w2s = 1.0f, w2s_inv = 1.0f;  // fp16 split: the layers' weight scales (powers of two) and their inverses
};
// fused decoder, stage B: y2 -> ConvT(64,32,s2)+ReLU -> ConvT(32,1,s1)+Sigmoid -> per-image reduction (+ image store)
struct DecBArgs {
    const float* y2;
    const float* w3; const float* b3;   // packed [9][1][8][64][4], bias [32]
    const float* w4; float b4;          // [9 taps][32 ch], scalar bias
    int rows;             // decoder rows (images) in this launch
    RowMask live;
    int m0;               // index of row 0 inside the [group][row] batch
    int rows_per_group;
    GroupMap gm;
    int reward0;          // groups with pidx == 0: 1 = reward log-likelihood, 0 = Bernoulli entropy sum (others: entropy)
    int store0;           // groups with pidx == 0 store their image at slot t*S + sample
    float* val;           // [batch] per-image pixel sum (entropy sum, or log-likelihood sum)
    int parts;            // 1 = one workgroup per image; 4 = four (small launches): quarter sums go to valq [batch][4]
    float* valq;
    float* po;            // [slots][rows_per_group][4096] stored images
    int reward_intent;    // 0 = the shipped port's NCHW-broadcast reward target, 1 = the upstream-intent variant (reward_term below)
    const void* w3b3 = nullptr;      // options mfma_bf16x3 / mfma_f16x2: ConvT3's weights as 16-bit planes [4 ks][9 taps][planes][64 lanes][8] (bf16x3.hip, k_dec_b_b3)
    int split = 0;                   // 1 = three bf16 planes, 2 = two fp16 planes
    float w3s = 1.0f, w3s_inv = 1.0f;      // fp16 split: the power of two the packed weights were scaled by, and its inverse
};
// fused encoder trunk: o [rows][64][64] -> conv1..conv4 (+ReLU) -> out [rows][576] in NHWC (p*64 + c) order
struct EncArgs {
    const float* o; float* out;
    const float* w1; const float* b1;   // conv1 [9 taps][32], [32]
    const float* w2; const float* b2;   // packed [9][1][4][64][4]
    const float* w3; const float* b3;   // packed [9][2][4][64][4]
    const float* w4; const float* b4;   // packed [9][4][4][64][4] (16x16x4 form)
    int rows;
    RowMask live;
};
void launch_enc_trunk(const EncArgs& a, hipStream_t st);
void launch_fc4(const GemmArgs& a, hipStream_t st);      // Linear(256,16384)+ReLU+Dropout with the batch tile staged in LDS
void launch_fc4_b3(const GemmArgs& a, hipStream_t st);   // the same on the bf16 pipe, operands split in three (opt-in experiment, bf16x3.hip)
float pack_dense_split(int mode, const float* W, const int* row_perm, int out, int in, uint16_t* dst);      // mode 1 = bf16 x 3, 2 = fp16 x 2; -> weight scale
void launch_dec_a_b3(const DecAArgs& a, hipStream_t st);  // k_dec_a's two layers on the bf16 pipe, every operand through LDS (opt-in experiment)
float pack_conv_split(int mode, const float* W_cicokk, int Cin, int Cout, uint16_t* dst);
void launch_dec_b_b3(const DecBArgs& a, hipStream_t st);  // k_dec_b4's layers with ConvT3 on the bf16 pipe (opt-in experiment)
float pack_convt3_split(int mode, const float* W_cicokk, uint16_t* dst);      // -> the weight scale (1 for the bf16 split)
int init_bf16x3_kernels();
void launch_dec_a(const DecAArgs& a, hipStream_t st);
void launch_dec_b(const DecBArgs& a, hipStream_t st);

int launch_dense(int MT, int NT, const GemmArgs& a, hipStream_t st);   // 0 = launched, 1 = unsupported tile shape

struct TransPostArgs {
    const float* tr;       // [2S][R][32] (mean 0..9, logvar 10..19); group order T1_0..T1_{S-1}, T2_0..T2_{S-1}
    const float* x;        // [R][16] current [pi(4) | s0(10) | 0 0]
    const float* eps_inj;  // nullable [3S][R][10]: T1_i, T2_j, D2B_j
    const float* given_ps1;// nullable [R][10]: trajectory mode, D1 input
    float* dec_in;         // [3S][R][16]: D1_i, D2A_j, D2B_j
    float* next_x;         // nullable [R][16]
    float* ps1_last;       // nullable [R][10]
    float* ps1_mean_last;  // nullable [R][10]
    int S, R, mean_mode, carry_mean;
    uint32_t k0, k1, stage, row_offset;
    int pi_dim;            // x rows are [pi (pi_dim) | s (10) | zeros]
    const int32_t* ids; int ids_div;            // optional row identities (GroupMap)
};
void launch_trans_post(const TransPostArgs& a, hipStream_t st);

struct TermsArgs {
    const float* val;      // [D][3S][R] decoder per-image pixel sums (D1_i reward log-lik, D2A_j / D2B_j entropy)
    const float* valq;     // nullable [D][3S][R][4]: the same as quarter sums (k_dec_b4 with four workgroups per image): val = (q0 + q1) + (q2 + q3)
    const float* tr;       // [D][2S][R][32]
    const float* enc;      // [D][S][R][32]
    int D, S, R;
    float reward_div;      // term0 per image = pixel sum / reward_div * 10 (mean over the counted pixels * 10, torchmodel.py:212: 4096, or the 192 bar
                           // pixels of the upstream-intent target); 0 = the sum form of the generic geometries
    float* G;              // [R] summed over stages
    float* terms;          // [3][R] summed over stages
    float* t2parts;        // nullable [2][R]: term2_1, term2_2 (last stage; diagnostics)
};
void launch_terms(const TermsArgs& a, hipStream_t st);
int init_small_kernels();       // per-device kernel attributes (dynamic LDS above 64 KiB); 0 = ok
int init_decoder_kernels();

// ---- device-resident MCTS tree (mcts.hip) --------------------------------------------------------
struct MctsTree {
    float* W; float* N; float* Qpi;   // [E][cap][A] edge statistics of mcts.py Node (W = -sum G, N visits, Qpi habit prior)
    int32_t* child;                   // [E][cap][A] node index of each child, -1 = not expanded
    float* S;                         // [E][cap][s_dim] latent state of every node
    int E, cap, A, s_dim;
};
void launch_mcts_select(const MctsTree& t, const uint8_t* active, float C, int use_prior, int max_depth, int32_t* path_nodes,
                        int32_t* path_act, int32_t* path_len, int32_t* leaf, float* leaf_s, float* leaf_s_rep, hipStream_t st);
void launch_mcts_expand(const MctsTree& t, int32_t* n_nodes, const int32_t* nodes, const uint8_t* mask, const float* G,
                        const float* ps_next, hipStream_t st);
void launch_mcts_backprop(const MctsTree& t, const int32_t* path_nodes, const int32_t* path_act, const int32_t* path_len,
                          const int32_t* leaf, const uint8_t* active, const float* sims, int R, const float* q0, int max_depth,
                          float* g_out, uint8_t* active_out, hipStream_t st);
void launch_mcts_stop(const MctsTree& t, uint8_t* active, int32_t* stop_at, int repeat, float threshold, int32_t* n_active, hipStream_t st);
struct MctsStepArgs {
    // back-propagation of the previous iteration (prev_path_len == nullptr: none); path_nodes / leaf still hold that iteration's selection
    const int32_t* prev_path_act; const int32_t* prev_path_len; const float* sims; int n_sims; const float* q0; float* prev_g_out; uint8_t* prev_active_out;
    // early stop of this iteration
    uint8_t* active; int32_t* stop_at; int repeat; float threshold; int32_t* n_active;
    // selection
    float C; int use_prior, max_depth;
    int32_t *path_nodes, *path_act, *path_len, *leaf; float *leaf_s, *leaf_s_rep;
    // expansion bookkeeping of the previous iteration's leaf (exp_n_nodes == nullptr: done by the caller with efe_mcts_expand)
    int32_t* exp_n_nodes = nullptr; const float* exp_G = nullptr; const float* exp_ps_next = nullptr;
};
void launch_mcts_step(const MctsTree& t, const MctsStepArgs& a, hipStream_t st);

// ---- fused small-MLP kernels (fused.hip) ---------------------------------------------------------------------
// weights packed for v_mfma_f32_16x16x4_f32: [16-feature tile][16-channel chunk][64 lanes][4], biases padded to the tile count
struct MlpW { const float4* w[4]; const float* b[4]; };
// the dense heads next to the conv stacks, one launch each (fused.hip k_head):
//   decoder head po_net.0/.3/.6 (torchmodel.py:107-115): s[16] -> 256 -> 256 -> 256, ReLU + MC-dropout each          -> Y [M][256]
//   encoder head qs_net.9/.12/.15/.18 (torchmodel.py:94-103): flat -> 256 -> 256 -> 256 (same) -> 20 (linear)       -> Y [M][32]
struct HeadArgs {
    MlpW W;                // packed for the 16x16x4 form (pack_linear16)
    int kc0;               // 16-channel chunks of the first layer's input: X rows are 16 * kc0 floats
    int nl;                // 3: Y = the third hidden layer [M][256];  4: + a linear layer of out_tiles 16-feature tiles, Y [M][16 * out_tiles]
    int out_tiles;
    uint32_t tag0;         // noise tag of the first layer (layer i uses tag0 + i)
    const float* X; float* Y;
    int M;
    uint32_t k0, k1;
    GroupMap gm; int rows_per_group; uint32_t row_offset; int m0;
};
void launch_head(const HeadArgs& a, hipStream_t st);

struct TransFusedArgs {
    MlpW W;                // ps_net.0 / .3 / .6 / .9
    const float* X;        // [rows][16] = [pi | s0 | 0 0]
    float* tr;             // [M][32]: mean 0..9, logvar 10..19
    int M, x_mod;          // x_mod > 0: input row = m % x_mod (every MC group reads the same rows)
    uint32_t k0, k1;
    GroupMap gm; int rows_per_group; uint32_t row_offset; int m0;
};
void launch_trans_fused(const TransFusedArgs& a, hipStream_t st);
struct SimChainArgs {
    MlpW W, H;             // transition net, habit net (qpi_net.0 / .2 / .4)
    const float* s0;       // [E][10] starting states
    int E, T, use_means;
    uint32_t k0, k1, stage, row_offset;
    const float* eps_inj;  // nullable [T][E][10]
    const float* u_inj;    // nullable [T][E]
    float *s0_traj, *ps1_traj, *mean_traj, *lv_traj;   // [E][T][10]
    float* pi0;            // [E][T][pi_dim] one-hot
    float* Qpi0;           // nullable [E][pi_dim]
    float* tr;             // nullable [2][E * T][32]: the trajectory core's transition rows (given T1 | T2 of the same input)
    int pi_dim;
    const int32_t* ids;                         // optional episode identities: episode slot e is episode ids[e] of the un-compacted batch
    // optional (one-episode decisions): a group of 8 episodes on EIGHT workgroups that split the two wide transition layers; exchange buffer
    // [groups][2][16][512] floats and self-resetting counters [groups][4] ints (zero at first use), both owned by the context
    float* xch = nullptr; int* sync = nullptr;
};
constexpr int SIM_FE = 8;                        // episodes per group of the simulation chain kernel
constexpr int SIM_MAX_SPLIT_GROUPS = 2;          // launches of up to 16 episodes use the split form
void launch_sim_chain(const SimChainArgs& a, hipStream_t st);
int init_fused_kernels();

// ---- geometry-generic convolution path (generic.hip) ---------------------------------------------
struct ConvGArgs {
    const float* in; float* out;      // NHWC activations
    const float* Wp; const float* bias; const float* zeros;     // packed [tap][32-feature tile][8-channel chunk][64 lanes][4]
    int n_img, Hin, Win, Cin;         // Cin padded to a multiple of 8
    int Hout, Wout, Cout, mtiles;
    int mode;                         // 0 Conv2d(k3,s2,p0)   1 ConvTranspose2d(k3,s1,p1)   2 ConvTranspose2d(k3,s2,p1,op1), sub-pixel form
    int relu;
    int ldo;                          // floats per output pixel
    RowMask live;                     // LDS-tiled ConvT kernels only (the decoder's layers)
};
void launch_conv_g(const ConvGArgs& a, hipStream_t st);
bool convt_p_ok(const ConvGArgs& a);                            // generic_dec.hip: the LDS-tiled ConvTranspose kernel takes this layer
void launch_convt_p(const ConvGArgs& a, hipStream_t st);
// ConvT(64,64,s1) + ReLU + ConvT(64,64,s2) + ReLU in one kernel, layer 1's output kept in LDS (generic_dec.hip: k_convt_12)
struct ConvT12Args {
    const float* in; float* out;        // [n][Hin * Win][64] -> [n][4 Hin * Win][64], NHWC
    const float* W1p; const float* b1;  // layer 1 as ConvGArgs::Wp / bias (packed [9][2][8][64][4])
    const float* W2p; const float* b2;  // layer 2
    int n_img, Hin, Win;
    RowMask live;
};
int launch_convt_12(const ConvT12Args& a, hipStream_t st);      // non-zero: outside the kernel's limits (launch the layers one by one)
// fused last two decoder layers of the generic path (generic_dec.hip): y2 [rows][Hin * Win][64] NHWC -> per-image sums (+ stored images)
struct DecBGArgs {
    const float* y2;
    const float* w3; const float* b3;   // packed [9][1][8][64][4], bias [32]
    const float* w4; float b4[4];       // [9 taps][32 ci][4 c], bias
    int rows, m0, rows_per_group, Hin, Win, C;
    int TH, RPa; unsigned magicWin;     // set by launch_dec_bg: input rows per strip, ring slots of the LDS strip, ceil(2^32 / Win)
    GroupMap gm;
    int reward0, store0, reward_intent;
    RowMask live;
    float* val;                         // [batch] per-image sums
    float* po;                          // [slots][rows_per_group][Hout * Wout][4] stored images (NHWC, channels padded to 4)
};
// LDS-tiled encoder Conv2d(k3, s2, p0) + ReLU of the generic path, layers 1 and 2 (generic_enc.hip)
struct ConvEArgs {
    const float* in;                    // layer 1: NHWC4 image [n][Hin * Win][4]; layer 2: [n][Hin * Win][32]
    float* out;                         // [n][Hout * Wout][32]
    const float* Wp; const float* bias; // layer 2: packed [9][1][4][64][4]; layer 1: [9 taps][64 lanes][2] = W[co][h][tap], W[co][2 + h][tap]
    int n_img, Hin, Win, Hout, Wout;
    int TY; unsigned magicWin, magicWout;       // set by launch_conv_e
    RowMask live;
};
int launch_conv_e(ConvEArgs a, int layer, hipStream_t st);     // non-zero: outside the kernel's limits (use k_conv_g)
// the two layers in one kernel, conv1's output kept in LDS (generic_enc.hip: k_conv_e12)
struct ConvE12Args {
    const float* in;                    // NHWC4 image [n][H0 * W0][4]
    float* out;                         // [n][H2 * W2][32]
    const float* W1p; const float* b1;  // layer 1 as ConvEArgs::Wp / bias of layer 1
    const float* W2p; const float* b2;  // layer 2 as ConvEArgs::Wp / bias of layer 2
    int n_img, H0, W0, H1, W1, H2, W2;
    int TY; unsigned magicW1, magicW2;          // set by launch_conv_e12
    RowMask live;
};
int launch_conv_e12(ConvE12Args a, hipStream_t st);             // non-zero: outside the kernel's limits (launch the layers one by one)
int init_generic_enc_kernels();
int launch_dec_bg(DecBGArgs a, hipStream_t st);                 // non-zero: geometry outside the kernel's limits
bool dec_bg_ok(int Hin, int Win, int C);                        // the fused kernel takes this geometry (decided before scratch is sized)
int init_generic_dec_kernels();
struct FinalGArgs {
    const float* y3;                  // [rows][H*W][32]
    const float* w; float b[4];       // [9 taps][32 ci][4 c], bias
    int rows, m0, rows_per_group, H, W, C;
    GroupMap gm;
    int reward0, store0, reward_intent;
    RowMask live;
    float* val;                       // [batch] per-image sums
    float* po;                        // [slots][rows_per_group][H*W][4] stored images (NHWC, channels padded to 4)
};
int launch_final_g(const FinalGArgs& a, hipStream_t st);       // non-zero: geometry outside the kernel's limits (W <= 128, C <= 3)
int init_generic_kernels();
constexpr int GEN_IMG_LD = 4;          // floats per pixel of the generic path's stored images (NHWC, C <= 3 channels padded to 4)
void launch_to_nhwc4(const float* in, float* out, long M, int HW, int C, hipStream_t st);       // NCHW -> NHWC4
void launch_nhwc4_to_8(const float* in, float* out, long n_pix, hipStream_t st);                // NHWC4 -> NHWC8 (k_conv_g's first layer contracts 8 input channels)
void launch_to_nchw(const float* in, float* out, long M, int HW, int C, hipStream_t st);
void launch_check_reward_g(const float* o, float* out, int M, int C, int H, int W, int intent, hipStream_t st);

void launch_pack_x(const float* pi, const float* s, float* x, int R, int pi_dim, int s_dim, hipStream_t st);
void launch_pad16(const float* s, float* x, int R, int s_dim, hipStream_t st);
void launch_root_post(const float* enc, const float* pi, const float* eps_inj, float* x, float* s_out, int R, int use_mean,
                      uint32_t k0, uint32_t k1, uint32_t pass, uint32_t sample, uint32_t stage, uint32_t row_offset, int pi_dim, hipStream_t st);
void launch_split_enc(const float* enc, float* mean, float* logvar, int R, hipStream_t st);
void launch_softmax4(const float* logits32, float* logits, float* q, float* logq, int R, int n, hipStream_t st);
void launch_mean_rows(const float* G, float* out, int E, int T, hipStream_t st);
void launch_fill_tr(const float* mean, const float* logvar, float* tr, int R, hipStream_t st);
void launch_posterior(const float* sumG, float* P, float* logP, int n_groups, int n, float temperature, hipStream_t st);

void launch_check_reward(const float* o, float* out, int M, int intent, hipStream_t st);
void launch_reparam(const float* mean, const float* logvar, const float* eps_inj, float* out, int M, int n, uint32_t k0, uint32_t k1,
                    uint32_t pass, uint32_t sample, uint32_t stage, uint32_t row_offset, hipStream_t st);
void launch_env_step(float* state, float* last_r, const int* actions, int* round_changed, int E, int repeats,
                     uint32_t k0, uint32_t k1, uint32_t stage, uint32_t game_offset, hipStream_t st);
void launch_env_new_image(float* state, int E, uint32_t k0, uint32_t k1, uint32_t stage, uint32_t game_offset, hipStream_t st);
void launch_env_reset(float* state, float* last_r, int E, uint32_t k0, uint32_t k1, uint32_t stage, uint32_t game_offset, hipStream_t st);
void launch_env_render(const float* state, const float* last_r, const unsigned char* imgs, long n_imgs, float* frames, int* err,
                       int E, hipStream_t st);

}  // namespace efe
