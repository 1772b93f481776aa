// Device-resident tree of the lock-step MCTS planner: the per-episode statistics of
// /root/reference/src/mcts.py `Node` (W, N, Qpi, children) live in [E][cap][A] device arrays, one thread walks /
// updates one episode's tree, so a planning iteration needs no host round trip between the engine calls.
//
// Every formula is evaluated in the reference's fp32 operation order (mcts.py:36-57, 82-99, 130-135):
//   Q = W / N;  Qn = (Q - min Q) / sum(Q - min Q)  (sum left to right);  score = Qn + C / N  (or Qn + (C * Qpi) / N, mcts.py:45)
// with torch's NaN rules (min and argmax propagate / prefer NaN, first index wins ties): an unvisited edge gives 0/0.
#include "kernels.h"

namespace efe {

// keeps hipcc from contracting  a * b + c  into an fma (the reference rounds the product first)
__device__ __forceinline__ float rounded(float x) { asm volatile("" : "+v"(x)); return x; }

__device__ __forceinline__ int argmax_nan_first(const float* v, int n) {          // torch.argmax: NaN is the maximum, first index wins
    int best = 0;
    for (int i = 1; i < n; ++i) {
        const bool nb = v[best] != v[best], ni = v[i] != v[i];
        if (nb) continue;
        if (ni || v[i] > v[best]) best = i;
    }
    return best;
}

// Tree-policy walk of ONE episode (mcts.py:49-62).  A level costs ONE memory latency: the W / N / Qpi / child rows of the current node are
// requested together, and "the node just reached has no children" (mcts.py:56) is read off the child row that the next level loads anyway.
// Entries of path_nodes / path_act at and beyond path_len are not written (every consumer stops at path_len).
__device__ __forceinline__ void mcts_select_one(const MctsTree& t, int e, const uint8_t* active, float C, int use_prior, int max_depth,
                                                int32_t* path_nodes, int32_t* path_act, int32_t* path_len, int32_t* leaf,
                                                float* leaf_s /*[E][s_dim]*/, float* leaf_s_rep /*[E*A][s_dim]*/) {
    const int A = t.A;
    int cur = 0, len = 0;
    if (active[e]) {
        for (int d = 0; d < max_depth; ++d) {
            const size_t o = ((size_t)e * t.cap + cur) * A;
            float w[8], n[8], qp[8]; int ch[8];
            for (int i = 0; i < A; ++i) { w[i] = t.W[o + i]; n[i] = t.N[o + i]; ch[i] = t.child[o + i]; qp[i] = use_prior ? t.Qpi[o + i] : 0.f; }
            if (d > 0 && ch[0] < 0) break;                           // the node reached by the previous level is a leaf
            float q[8], sc[8];
            float qmin = 0.f; bool nan_min = false;
            for (int i = 0; i < A; ++i) {
                q[i] = w[i] / n[i];
                if (q[i] != q[i]) nan_min = true;
                if (i == 0 || q[i] < qmin) qmin = q[i];
            }
            if (nan_min) qmin = __builtin_nanf("");
            float sum = 0.f;
            for (int i = 0; i < A; ++i) { q[i] = q[i] - qmin; sum = (i == 0) ? q[i] : sum + q[i]; }
            for (int i = 0; i < A; ++i) {
                // mcts.py:45-47: `C * Qpi * 1.0 / N` evaluates left to right as ((C * Qpi) * 1.0) / N; without the prior (C * 1.0) / N
                const float bonus = use_prior ? rounded(C * qp[i]) / n[i] : C / n[i];
                sc[i] = rounded(q[i] / sum) + bonus;
            }
            const int a = argmax_nan_first(sc, A);
            path_nodes[(size_t)e * max_depth + d] = cur;
            path_act[(size_t)e * max_depth + d] = a;
            len = d + 1;
            cur = ch[a];
        }
    }
    path_len[e] = len;
    leaf[e] = cur;
    const float* s = t.S + ((size_t)e * t.cap + cur) * t.s_dim;
    for (int k = 0; k < t.s_dim; ++k) {
        const float v = s[k];
        leaf_s[(size_t)e * t.s_dim + k] = v;
        for (int a = 0; a < A; ++a) leaf_s_rep[((size_t)e * A + a) * t.s_dim + k] = v;
    }
}
__global__ void k_mcts_select(const MctsTree t, const uint8_t* active, float C, int use_prior, int max_depth,
                              int32_t* path_nodes, int32_t* path_act, int32_t* path_len, int32_t* leaf,
                              float* leaf_s, float* leaf_s_rep) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t.E) return;
    mcts_select_one(t, e, active, C, use_prior, max_depth, path_nodes, path_act, path_len, leaf, leaf_s, leaf_s_rep);
}

// Node.expand bookkeeping (mcts.py:64-86): W -= G, N += 1, pi_dim children with the predicted states
__device__ __forceinline__ void mcts_expand_one(const MctsTree& t, int e, int32_t* n_nodes, const int32_t* nodes, const uint8_t* mask, const float* G,
                                                const float* ps_next) {
    if (!mask[e]) return;
    const int A = t.A, n = nodes[e];
    const size_t o = ((size_t)e * t.cap + n) * A;
    const int base = n_nodes[e];
    for (int a = 0; a < A; ++a) {
        t.W[o + a] -= G[(size_t)e * A + a];
        t.N[o + a] += 1.0f;
        t.child[o + a] = base + a;
        float* sd = t.S + ((size_t)e * t.cap + base + a) * t.s_dim;
        const float* ss = ps_next + ((size_t)e * A + a) * t.s_dim;
        for (int k = 0; k < t.s_dim; ++k) sd[k] = ss[k];
    }
    n_nodes[e] = base + A;
}
__global__ void k_mcts_expand(const MctsTree t, int32_t* n_nodes, const int32_t* nodes, const uint8_t* mask, const float* G,
                              const float* ps_next) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t.E) return;
    mcts_expand_one(t, e, n_nodes, nodes, mask, G, ps_next);
}

// after the simulations: habit prior of the leaf, g = mean of the simulated G (left to right), back-propagation along the
// selected path (mcts.py:91-99, 186-191), and the iteration's history row
__device__ __forceinline__ void mcts_backprop_one(const MctsTree& t, int e, const int32_t* path_nodes, const int32_t* path_act, const int32_t* path_len,
                                                  const int32_t* leaf, const uint8_t* active, const float* sims /*[R][E]*/, int R, const float* q0 /*[E][A]*/,
                                                  int max_depth, float* g_out /*[E]*/, uint8_t* active_out /*[E]*/) {
    float g = 0.f;
    for (int r = 0; r < R; ++r) g = (r == 0) ? sims[(size_t)r * t.E + e] : g + sims[(size_t)r * t.E + e];
    g = g / (float)R;
    g_out[e] = g;
    active_out[e] = active[e];
    if (!active[e]) return;
    const int A = t.A;
    const size_t ol = ((size_t)e * t.cap + leaf[e]) * A;
    for (int a = 0; a < A; ++a) t.Qpi[ol + a] = q0[(size_t)e * A + a];
    // (a path visits every node once: its entries are independent, requested back to back)
    const int L = path_len[e];
    for (int d0 = 0; d0 < L; d0 += 8) {
        size_t o[8]; float w[8], n[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = d0 + i < L ? d0 + i : d0;                  // (entries past the path re-read a valid one and store nothing)
            o[i] = ((size_t)e * t.cap + path_nodes[(size_t)e * max_depth + d]) * A + path_act[(size_t)e * max_depth + d];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { w[i] = t.W[o[i]]; n[i] = t.N[o[i]]; }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (d0 + i < L) { t.W[o[i]] = w[i] - g; t.N[o[i]] = n[i] + 1.0f; }
    }
}
__global__ void k_mcts_backprop(const MctsTree t, const int32_t* path_nodes, const int32_t* path_act, const int32_t* path_len,
                                const int32_t* leaf, const uint8_t* active, const float* sims, int R, const float* q0,
                                int max_depth, float* g_out, uint8_t* active_out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t.E) return;
    mcts_backprop_one(t, e, path_nodes, path_act, path_len, leaf, active, sims, R, q0, max_depth, g_out, active_out);
}

// early stop (mcts.py:130-131, 176): an episode is done when max(N/sum N) - mean(N/sum N) at the root exceeds the threshold
__device__ __forceinline__ void mcts_stop_one(const MctsTree& t, int e, uint8_t* active, int32_t* stop_at, int repeat, float threshold,
                                              int32_t* n_active) {
    if (!active[e]) return;
    const int A = t.A;
    const float* n = t.N + (size_t)e * t.cap * A;            // root = node 0
    float sum = 0.f;
    for (int a = 0; a < A; ++a) sum = (a == 0) ? n[a] : sum + n[a];
    float dist[8], dsum = 0.f, dmax = 0.f; bool isn = false;
    for (int a = 0; a < A; ++a) {
        dist[a] = n[a] / sum;
        if (dist[a] != dist[a]) isn = true;
        if (a == 0 || dist[a] > dmax) dmax = dist[a];
        dsum = (a == 0) ? dist[a] : dsum + dist[a];
    }
    if (isn) dmax = __builtin_nanf("");
    const float crit = dmax - dsum / (float)A;
    if (crit > threshold) { active[e] = 0; stop_at[e] = repeat; }
    else atomicAdd(n_active, 1);
}
__global__ void k_mcts_stop(const MctsTree t, uint8_t* active, int32_t* stop_at, int repeat, float threshold, int32_t* n_active) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t.E) return;
    mcts_stop_one(t, e, active, stop_at, repeat, threshold, n_active);
}
// One launch for the tree work between two iterations' engine calls: expansion bookkeeping and back-propagation of the previous iteration
// (skipped for the first), early-stop test, selection of the next leaf -- the three one-thread-per-episode kernels above, in that order, per episode.
// n_active is a zero-initialised word of its own per iteration (no memset launch).
__global__ void k_mcts_step(const MctsTree t, MctsStepArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t.E) return;
    if (a.exp_n_nodes) mcts_expand_one(t, e, a.exp_n_nodes, a.leaf, a.active, a.exp_G, a.exp_ps_next);       // the previous iteration's leaf (a.leaf is rewritten by the selection below)
    if (a.prev_path_len)
        mcts_backprop_one(t, e, a.path_nodes, a.prev_path_act, a.prev_path_len, a.leaf, a.active, a.sims, a.n_sims, a.q0, a.max_depth, a.prev_g_out,
                          a.prev_active_out);
    mcts_stop_one(t, e, a.active, a.stop_at, a.repeat, a.threshold, a.n_active);
    mcts_select_one(t, e, a.active, a.C, a.use_prior, a.max_depth, a.path_nodes, a.path_act, a.path_len, a.leaf, a.leaf_s, a.leaf_s_rep);
}
void launch_mcts_step(const MctsTree& t, const MctsStepArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(k_mcts_step, dim3((t.E + 63) / 64), dim3(64), 0, st, t, a);
}

void launch_mcts_select(const MctsTree& t, const uint8_t* active, float C, int use_prior, int max_depth, int32_t* path_nodes,
                        int32_t* path_act, int32_t* path_len, int32_t* leaf, float* leaf_s, float* leaf_s_rep, hipStream_t st) {
    hipLaunchKernelGGL(k_mcts_select, dim3((t.E + 63) / 64), dim3(64), 0, st, t, active, C, use_prior, max_depth, path_nodes, path_act,
                       path_len, leaf, leaf_s, leaf_s_rep);
}
void launch_mcts_expand(const MctsTree& t, int32_t* n_nodes, const int32_t* nodes, const uint8_t* mask, const float* G,
                        const float* ps_next, hipStream_t st) {
    hipLaunchKernelGGL(k_mcts_expand, dim3((t.E + 63) / 64), dim3(64), 0, st, t, n_nodes, nodes, mask, G, ps_next);
}
void launch_mcts_backprop(const MctsTree& t, const int32_t* path_nodes, const int32_t* path_act, const int32_t* path_len,
                          const int32_t* leaf, const uint8_t* active, const float* sims, int R, const float* q0, int max_depth,
                          float* g_out, uint8_t* active_out, hipStream_t st) {
    hipLaunchKernelGGL(k_mcts_backprop, dim3((t.E + 63) / 64), dim3(64), 0, st, t, path_nodes, path_act, path_len, leaf, active, sims, R,
                       q0, max_depth, g_out, active_out);
}
void launch_mcts_stop(const MctsTree& t, uint8_t* active, int32_t* stop_at, int repeat, float threshold, int32_t* n_active, hipStream_t st) {
    (void)hipMemsetAsync(n_active, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(k_mcts_stop, dim3((t.E + 63) / 64), dim3(64), 0, st, t, active, stop_at, repeat, threshold, n_active);
}

}  // namespace efe
