"""Single-episode MCTS planner over the engine: same public API as /root/reference/src/mcts.py
(`Node`, `MCTS_Params`, `active_inference_mcts`) so existing callers keep working.  The tree is tiny
host-side bookkeeping (pi_dim floats per node); every network evaluation goes to the HIP engine through
`model.calculate_G[_mean]` / `model.mcts_step_simulate`.
"""
import numpy as np
import torch

_OPPOSITE = {4: {(0, 1), (1, 0), (2, 3), (3, 2)}, 3: {(1, 2), (2, 1)}}


def calc_threshold(P, axis):
    """max - mean of a distribution (mcts.py:130-131)"""
    return torch.max(P, dim=axis).values - torch.mean(P, dim=axis)


def normalization(x, tau=1):
    """x / sum(x) (mcts.py:133-135)"""
    return x / x.sum(dim=0)


class MCTS_Params:
    """Planner knobs with the reference's names and defaults (mcts.py:137-148)."""

    def __init__(self):
        self.C = 1.0
        self.threshold = 0.5
        self.repeats = 300
        self.simulation_repeats = 1
        self.simulation_depth = 3
        self.use_habit = False
        self.use_means = True
        self.verbose = False
        self.method = 'ai'
        self.using_prior_for_exploration = False
        self.samples = 1          # extension: MC samples per expansion (reference always expands with 1)


class Node:
    """One tree node = one latent state, replicated pi_dim times so an expansion is a pi_dim-row batch
    (mcts.py:11-34)."""
    _next_id = 0

    def __init__(self, s, model, C, pi_dim=4, verbose=False, using_prior_for_exploration=False):
        self.pi_dim = pi_dim
        self.s = s.reshape(1, -1).repeat(pi_dim, 1)
        self.model = model
        self.verbose = verbose
        self.using_prior_for_exploration = using_prior_for_exploration
        self.visited = False
        self.NODE_ID = Node._next_id
        Node._next_id += 1
        self.W = torch.zeros(pi_dim)       # accumulated -G per edge
        self.N = torch.zeros(pi_dim)       # visit count per edge
        self.Qpi = torch.zeros(pi_dim)     # habit prior
        self.children_nodes = [None] * pi_dim
        self.C = C
        self.in_progress = -1

    def Q(self):
        return self.W / self.N

    def probs_for_selection(self):
        """Q normalised to a distribution plus the C/N exploration bonus (mcts.py:39-47)."""
        q = self.Q()
        q = q - q.min()
        q = q / q.sum()
        if self.using_prior_for_exploration:
            return q + self.C * self.Qpi * 1.0 / self.N       # ((C * Qpi) * 1.0) / N, the reference's fp32 order (mcts.py:45)
        return q + self.C * 1.0 / self.N

    def _pick(self, scores, deterministic):
        return int(torch.argmax(scores)) if deterministic else int(torch.multinomial(scores, 1))

    def select(self, deterministic=True):
        """Walk down to a leaf following the tree policy (mcts.py:49-62)."""
        path, actions = [], []
        node = self
        while True:
            node.in_progress = node._pick(node.probs_for_selection(), deterministic)
            actions.append(node.in_progress)
            node = node.children_nodes[node.in_progress]
            path.append(node)
            if any(c is None for c in node.children_nodes):
                return path, actions

    def expand(self, use_means=False, samples=1):
        """Evaluate all actions of a leaf in one engine call (mcts.py:64-86)."""
        if self.pi_dim == 4:
            pi_hot = self.model.pi_one_hot
        elif self.pi_dim == 3:
            pi_hot = self.model.pi_one_hot_3
        else:
            raise ValueError(f'unsupported pi_dim {self.pi_dim}')
        if use_means:
            G, _, ps_next, _ = self.model.calculate_G_mean(self.s, pi_hot)
        else:
            G, _, ps_next, _, _ = self.model.calculate_G(self.s, pi_hot, samples=samples)
        self.W -= G.detach().to('cpu')
        self.N += 1.0
        for a in range(self.pi_dim):
            self.children_nodes[a] = Node(ps_next[a], self.model, self.C, self.pi_dim,
                                          using_prior_for_exploration=self.using_prior_for_exploration)

    def backpropagate(self, path, G):
        """Credit -G to the edge taken at every node on the path (mcts.py:88-96)."""
        for node in path:
            if node.in_progress < 0:
                raise ValueError('back-propagation through a node with no edge in progress')
            node.W[node.in_progress] -= G
            node.N[node.in_progress] += 1
            node.in_progress = -2

    def action_selection(self, deterministic=True):
        """Most-visited path to a leaf, minus its last action, with back-and-forth pairs removed
        (mcts.py:98-128)."""
        visited = []
        node = self
        while True:
            a = node._pick(node.N if deterministic else node.N.float(), deterministic)
            visited.append(a)
            node = node.children_nodes[a]
            if any(c is None for c in node.children_nodes):
                break
        if self.pi_dim not in _OPPOSITE:
            raise ValueError(f'unknown pi_dim {self.pi_dim}')
        cancel = _OPPOSITE[self.pi_dim]
        trimmed, i = [], 0
        while i < len(visited) - 1:
            if (visited[i], visited[i + 1]) in cancel:
                i += 2
            else:
                trimmed.append(visited[i])
                i += 1
        return trimmed


def _frames_nchw(model, frames, n, o_shape):
    """[n, *o_shape] frames -> NCHW [n, C, R, R] as the engine takes them.  One channel: HWC (64, 64, 1) and CHW (1, 64, 64) are the
    same memory (the reference reshapes, mcts.py:158).  Several channels: (C, R, R) is taken as is, (R, R, C) is permuted --
    a plain reshape would scramble the channels -- and anything else is an error."""
    Cc, R = model.colour_channels, model.resolution
    fr = torch.as_tensor(frames)
    o_shape = tuple(int(v) for v in o_shape)
    if Cc == 1:
        if o_shape not in ((R, R, 1), (1, R, R), (R, R)):
            raise ValueError(f'o_shape {o_shape} does not match a 1 x {R} x {R} observation')
        return fr.reshape(n, 1, R, R)
    if o_shape == (Cc, R, R):
        return fr.reshape(n, Cc, R, R)
    if o_shape == (R, R, Cc):
        return fr.reshape(n, R, R, Cc).permute(0, 3, 1, 2).contiguous()
    raise ValueError(f'o_shape {o_shape}: a {Cc}-channel model takes ({Cc}, {R}, {R}) (NCHW) or ({R}, {R}, {Cc}) (HWC) frames')


def active_inference_mcts(model, frame, params, o_shape=(64, 64, 1)):
    """One planning decision (mcts.py:150-195) -> (path, repeats_done, states_explored, all_paths, all_paths_G).
    The decision runs on the device-resident planner (BatchedMCTS with one episode: tree statistics, selection, back-propagation and
    the early stop are kernels, the simulation runs beside the expansion on a second stream) -- per decision the same draws and the same results as
    the host-side Node tree below, which `params.host_tree = True` selects (it is what Node.expand / select / backpropagate users get).  The
    device planner reserves the noise stages of a whole decision up front, the host tree takes them as it goes: after a decision that ended
    early (habit shortcut, early stop) the NEXT decision on the same model draws from different stages in the two modes.  With the default
    parameters the planner also creates (once per weight version, cached on the model) a replica engine context for the second stream:
    INTEGRATION.md section 4; `overlap_simulate = False` avoids it."""
    if not getattr(params, 'host_tree', False) and not (frame is None or (isinstance(frame, (list, tuple)) and len(frame) == 0)):
        out, _ = active_inference_mcts_batch(model, torch.as_tensor(frame)[None], params, o_shape=o_shape)
        return out[0]
    prev = torch.get_num_threads()
    torch.set_num_threads(1)        # pi_dim-sized host tensors: keep torch's intra-op pool out of it
    try:
        return _mcts_one(model, frame, params, o_shape)
    finally:
        torch.set_num_threads(prev)


def _mcts_one(model, frame, params, o_shape):
    states_explored, all_paths, all_paths_G = 0, [], []
    if frame is None or (isinstance(frame, (list, tuple)) and len(frame) == 0):
        return [0], 0, states_explored, all_paths, all_paths_G

    qs0_mean, _ = model.model_down.encoder(_frames_nchw(model, frame, 1, o_shape))
    root = Node(qs0_mean[0], model, params.C, model.pi_dim, using_prior_for_exploration=params.using_prior_for_exploration)
    root.Qpi = model.model_top.encode_s(qs0_mean)[1][0].to('cpu')

    if params.use_habit and calc_threshold(root.Qpi, axis=0) > params.threshold:
        return [int(torch.multinomial(root.Qpi, 1))], 0, states_explored, all_paths, all_paths_G

    samples = getattr(params, 'samples', 1)
    root.expand(use_means=params.use_means, samples=samples)
    for repeat in range(params.repeats):
        if calc_threshold(normalization(root.N), axis=0) > params.threshold:
            return root.action_selection(deterministic=True), repeat, states_explored, all_paths, all_paths_G
        path, actions_path = root.select(deterministic=True)
        leaf = path[-1]
        leaf.expand(use_means=params.use_means, samples=samples)
        sims = torch.zeros(params.simulation_repeats)
        for k in range(params.simulation_repeats):
            states_explored += params.simulation_depth
            sims[k], _, qpi = model.mcts_step_simulate(leaf.s[0], params.simulation_depth, use_means=False)
            leaf.Qpi = qpi.to('cpu')
        leaf.backpropagate([root] + path[:-1], sims.mean())
        all_paths.append(actions_path)
        all_paths_G.append(sims.mean().item())
    return root.action_selection(deterministic=True), params.repeats, states_explored, all_paths, all_paths_G


# ---------------------------------------------------------------------------------------------------------
# Lock-step planner over E independent episodes: the same algorithm as active_inference_mcts,
# but every tree takes its select / expand / simulate / back-propagate step together, so one expansion is ONE
# engine call over E x pi_dim rows (x MC samples) and one simulation is ONE call over E episodes.
# Tree statistics are [E, nodes, pi_dim] host tensors; node states stay on the device.
# ---------------------------------------------------------------------------------------------------------
def _trim_path(visited, pi_dim):
    cancel = _OPPOSITE[pi_dim]
    trimmed, i = [], 0
    while i < len(visited) - 1:
        if (visited[i], visited[i + 1]) in cancel:
            i += 2
        else:
            trimmed.append(visited[i])
            i += 1
    return trimmed


class BatchedMCTS:
    """Lock-step planner over E episodes.  The tree statistics of every episode (mcts.py Node: W, N, Qpi,
    children, node states) are device-resident [E, nodes, pi_dim] arrays; selection, expansion bookkeeping, back-propagation
    and the early-stop test are one-thread-per-episode kernels behind the C ABI (`efe_mcts_*`, csrc/mcts.hip), so an
    iteration is a stream of launches with no host round trip between the engine calls.  The host only reads the number of
    still-active episodes (one int per iteration) and, at the end, the history needed for the reference's return tuple."""

    def __init__(self, model, n_episodes, params, episode_offset=0):
        import ctypes as C
        import weakref
        from . import _lib
        # (a weak reference: planner objects are cached ON the model, a strong one would make a cycle and defer the engine context's
        # release to the garbage collector)
        self._model_ref = weakref.ref(model)
        import copy
        # (a private copy: planners are cached by the VALUE of the parameters, a caller that later mutates its object must not change a
        # cached planner whose buffers were sized for the old values)
        self.E, self.p = int(n_episodes), copy.copy(params)
        if getattr(params, 'use_graph', False):
            # (removed with ABI 5: the hipGraph replay of an iteration.  Said once, loudly, instead of silently ignoring the attribute)
            import warnings
            warnings.warn('MCTS_Params.use_graph is no longer supported (removed with engine ABI 5) and is ignored: the planner launches its '
                          'iterations directly (INTEGRATION.md section 4)', RuntimeWarning, stacklevel=3)
        self.pi_dim = A = model.pi_dim
        self.ep0 = int(episode_offset)
        self.cap = cap = 1 + A * (params.repeats + 2)
        self.max_depth = params.repeats + 2
        E, dev = self.E, model.device
        self.W = torch.zeros(E, cap, A, device=dev)
        self.N = torch.zeros(E, cap, A, device=dev)
        self.Qpi = torch.zeros(E, cap, A, device=dev)
        self.child = torch.full((E, cap, A), -1, dtype=torch.int32, device=dev)
        self.S = torch.zeros(E, cap, model.s_dim, device=dev)
        self.n_nodes = torch.ones(E, dtype=torch.int32, device=dev)
        self._tree = _lib.EfeMctsTree(self.W.data_ptr(), self.N.data_ptr(), self.Qpi.data_ptr(), self.child.data_ptr(),
                                      self.S.data_ptr(), E, cap, A, model.s_dim)
        self._C = C
        self.pi_hot = (model.pi_one_hot if A == 4 else model.pi_one_hot_3).repeat(E, 1)
        # per-iteration scratch + history (device)
        R = max(1, params.repeats)
        self.path_nodes = torch.zeros(E, self.max_depth, dtype=torch.int32, device=dev)
        self.H_act = torch.zeros(R, E, self.max_depth, dtype=torch.int32, device=dev)
        self.H_len = torch.zeros(R, E, dtype=torch.int32, device=dev)
        self.H_g = torch.zeros(R, E, device=dev)
        self.H_active = torch.zeros(R, E, dtype=torch.uint8, device=dev)
        self.leaf = torch.zeros(E, dtype=torch.int32, device=dev)
        self.leaf_s = torch.zeros(E, model.s_dim, device=dev)
        self.leaf_rep = torch.zeros(E * A, model.s_dim, device=dev)
        self.sims = torch.zeros(max(1, params.simulation_repeats), E, device=dev)
        self.stop_at = torch.full((E,), -1, dtype=torch.int32, device=dev)
        self.n_active = torch.zeros(1, dtype=torch.int32, device=dev)
        self.n_active_it = torch.zeros(R + 1, dtype=torch.int32, device=dev)       # efe_mcts_step: one word per iteration
        self.q0 = torch.zeros(E, A, device=dev)
        self.active = torch.zeros(E, dtype=torch.uint8, device=dev)
        self.root_nodes = torch.zeros(E, dtype=torch.int32, device=dev)
        # expansion / simulation results of a COMPACTED call (only the live episodes, efe_rows.ids) are scattered into these full-size rows
        self.G_full = torch.zeros(E * A, device=dev)
        self.ps_full = torch.zeros(E * A, model.s_dim, device=dev)
        self.h_active = torch.zeros(4, E, dtype=torch.uint8).pin_memory()      # snapshots of `active` for the lagged host check
        self.ev_snap = [torch.cuda.Event() for _ in range(4)]
        self.h_ids = torch.zeros(4, E, dtype=torch.int64).pin_memory()
        self._sim_out = None
        self._rows_cache = {}
        self._ids = None                    # (int32 device tensor of live episode indices, the same as a host list, int64 copy for torch indexing)
        # The simulation of an iteration (habit rollout from the leaf + G over its trajectory: ~1 ms of small launches) is
        # independent of the expansion of the same leaf (~6 ms of large ones): with more than a few episodes it runs on a second
        # stream through a replica context, so its launch-bound chain hides under the expansion's MFMA-bound kernels.
        self.overlap = E >= int(getattr(params, 'overlap_min_episodes', 1)) and getattr(params, 'overlap_simulate', True)
        if self.overlap:
            self.sim_model = model.cached_replica()
            self.sim_stream = torch.cuda.Stream(device=dev)
            self.ev_sel = torch.cuda.Event()
            self.ev_sim = torch.cuda.Event()

    @property
    def model(self):
        return self._model_ref()

    def _call(self, fn, *args):
        e = self.model._ready()
        e.check(fn(e.ctx, self._C.byref(self._tree), *args, e.stream()))

    @staticmethod
    def _p(t):
        import ctypes as C
        return C.c_void_p(t.data_ptr())

    def _rows(self, mask, rows_per_entry):
        """the row set of an engine call of this iteration: the liveness mask (early-stopped / habit-decided episodes are skipped on the
        device) and, once episodes have been lost, only the live ones as a dense batch (self._ids, refreshed by _compact)"""
        from .model import Rows
        if mask is None and self._ids is None:
            return None
        # one Rows object per (mask, rows_per_entry) and compaction: built where the batch is compacted, reused by every call until the next one
        key = (None if mask is None else mask.data_ptr(), rows_per_entry)
        r = self._rows_cache.get(key)
        if r is None:
            ids, ids_host = (self._ids[0], self._ids[1]) if self._ids is not None else (None, None)
            r = self._rows_cache[key] = Rows(mask=mask, ids=ids, rows_per_entry=rows_per_entry, ids_host=ids_host, n_total=self.E)
        return r

    def _compact(self, n_live):
        """gather the live episodes into a dense batch for the following iterations (called where the host has just read the active count:
        no extra synchronisation; the lagged check uses _compact_host).  The dense layers, which cannot skip single rows, shrink too."""
        cur = self.E if self._ids is None else len(self._ids[1])
        if n_live <= 0 or n_live > (1.0 - float(getattr(self.p, 'compact_min_dead', 0.03))) * cur:
            return
        idx = torch.nonzero(self.active).flatten()
        self._ids = (idx.to(torch.int32).contiguous(), idx.cpu().tolist(), idx)
        self._rows_cache = {}

    def _compact_host(self, snap, n_live):
        """the same from a host snapshot of `active` (the lagged check): no device synchronisation at all"""
        cur = self.E if self._ids is None else len(self._ids[1])
        if n_live <= 0 or n_live > (1.0 - float(getattr(self.p, 'compact_min_dead', 0.03))) * cur:
            return
        ids = np.flatnonzero(snap)
        self._h_ids_k = (getattr(self, '_h_ids_k', -1) + 1) % self.h_ids.shape[0]           # (pinned staging, asynchronous upload: a pageable
        stage = self.h_ids[self._h_ids_k]                                                   # source would synchronise the whole stream)
        stage[:len(ids)] = torch.from_numpy(ids)
        idx = stage[:len(ids)].to(self.active.device, non_blocking=True)
        self._ids = (idx.to(torch.int32).contiguous(), ids.tolist(), idx)
        self._rows_cache = {}

    def _expand(self, nodes, mask, states_rep, stage=None, use_mask=True, defer=False):
        """ONE engine call over E x pi_dim rows (Node.expand, mcts.py:64-86); tree bookkeeping only where mask[e].  defer: the bookkeeping
        (W -= G, N += 1, children, their states) is left to the NEXT efe_mcts_step launch -> returns the (G, ps_next) tensors it needs"""
        m, p_, A = self.model, self._p, self.pi_dim
        ro = self.ep0 * A
        rows = self._rows(mask if use_mask else None, A)
        pi_hot = self.pi_hot
        if self._ids is not None:
            idx = self._ids[2]
            states_rep = states_rep.view(self.E, A, -1).index_select(0, idx).reshape(idx.numel() * A, -1)
            pi_hot = self.pi_hot[:idx.numel() * A]
        if self.p.use_means:
            G, _, ps_next, _ = m.calculate_G_mean(states_rep, pi_hot, row_offset=ro, stage=stage, rows=rows)
        else:
            G, _, ps_next, _, _ = m.calculate_G(states_rep, pi_hot, samples=getattr(self.p, 'samples', 1), row_offset=ro, stage=stage, rows=rows)
        if self._ids is not None:
            idx = self._ids[2]
            self.G_full.view(self.E, A).index_copy_(0, idx, G.view(-1, A))
            self.ps_full.view(self.E, A, -1).index_copy_(0, idx, ps_next.view(idx.numel(), A, -1))
            G, ps_next = self.G_full, self.ps_full
        G, ps_next = G.contiguous(), ps_next.contiguous()
        if defer:
            return G, ps_next
        self._call(m._engine.lib.efe_mcts_expand, p_(self.n_nodes), p_(nodes), p_(mask), p_(G), p_(ps_next))
        return None

    def _simulate(self, model, mask, stage_of, direct=False):
        """the iteration's simulations from the selected leaves (mcts.py:186-189) -> self.sims, self.q0 (direct: with one simulation per
        iteration and an un-compacted batch the engine's own output tensors are handed to the back-propagation: self._sim_out, no copies)"""
        p = self.p
        rows = self._rows(mask, 1)
        leaf_s = self.leaf_s if self._ids is None else self.leaf_s.index_select(0, self._ids[2])
        self._sim_out = None
        for r in range(p.simulation_repeats):
            G, _, q0 = model.simulate_batch(leaf_s, p.simulation_depth, use_means=False, row_offset=self.ep0, stage=stage_of(r), rows=rows)
            if direct and p.simulation_repeats == 1 and self._ids is None:
                self._sim_out = (G, q0)          # (kept until the next iteration's simulation, which starts behind this one's back-propagation)
            elif self._ids is None:
                self.sims[r].copy_(G)
                self.q0.copy_(q0)
            else:
                self.sims[r].index_copy_(0, self._ids[2], G)
                self.q0.index_copy_(0, self._ids[2], q0)

    def action_selection(self, e, N=None, child=None):
        N = self.N.cpu().numpy() if N is None else np.asarray(N)
        child = self.child.cpu().numpy() if child is None else np.asarray(child)
        visited, node = [], 0
        while True:
            a = int(np.argmax(N[e, node]))
            visited.append(a)
            node = int(child[e, node, a])
            if child[e, node, 0] < 0:
                break
        return _trim_path(visited, self.pi_dim)

    def reset(self):
        """fresh trees (a planner object, with its buffers, is reused across decisions)"""
        self.W.zero_(); self.N.zero_(); self.Qpi.zero_()
        self.child.fill_(-1); self.n_nodes.fill_(1); self.stop_at.fill_(-1)

    def run(self, frames, o_shape=(64, 64, 1)):
        # the little host-side work left (habit shortcut, final path read-out) uses tiny tensors: one intra-op thread
        prev = torch.get_num_threads()
        torch.set_num_threads(1)
        if self.overlap:
            self.sim_model = self.model.cached_replica()       # same object unless the weights changed; refreshed noise settings
        try:
            return self._run(frames, o_shape)
        finally:
            torch.set_num_threads(prev)

    def _run(self, frames, o_shape):
        m, E, p, A, p_ = self.model, self.E, self.p, self.pi_dim, self._p
        lib = m._engine.lib
        res = [None] * E
        qs0_mean, _ = m.model_down.encoder(_frames_nchw(m, frames, E, o_shape), row_offset=self.ep0)
        self.S[:, 0] = qs0_mean
        q_root = m.model_top.encode_s(qs0_mean)[1]
        self.Qpi[:, 0] = q_root
        active_h = torch.ones(E, dtype=torch.bool)
        if p.use_habit:
            q_cpu = q_root.to('cpu')
            for e in range(E):
                if calc_threshold(q_cpu[e], axis=0) > p.threshold:
                    res[e] = ([int(torch.multinomial(q_cpu[e], 1))], 0, 0, [], [])
                    active_h[e] = False
        # noise stages in the reference's call order (the root expansion takes one stage, then per iteration the expansion one and each
        # simulation one), reserved up front and indexed by the iteration: what a later call draws does not depend on E, on which episodes
        # share this planner / rank, or on when the loop ended -- also when every episode is decided by the habit shortcut below
        per_it = 1 + p.simulation_repeats
        st_root = m._take_stage(None, 1)
        st0 = m._take_stage(None, p.repeats * per_it)
        if not bool(active_h.any()):        # every episode was decided by the habit shortcut (mcts.py:166-169): nothing to plan
            return res
        self.active.copy_(active_h.to(torch.uint8))
        active = self.active
        # early-stopped (and habit-decided) episodes stop costing flops: every engine call of the loop carries `active` as its row mask
        # (efe_rows.mask: the per-image kernels read it on the device and skip dead rows; efe_mcts_stop clears entries as the loop runs, no
        # host round trip involved), and where the host learns the active count anyway the live episodes are compacted into a dense batch
        skip = bool(getattr(p, 'skip_stopped', True))
        compact = skip and bool(getattr(p, 'compact_stopped', True))
        self._ids, self._rows_cache = None, {}
        if compact and not bool(active_h.all()):
            self._compact(int(active_h.sum()))
        self._expand(self.root_nodes, active, self.S[:, 0].repeat_interleave(A, dim=0).contiguous(), stage=st_root, use_mask=skip)
        n_iter = 0
        # The per-episode early stop (mcts.py:176) is applied on the device every iteration (stopped episodes are masked out of
        # every tree update); the host only needs to know when ALL episodes have stopped, to end the loop early.  It looks at
        # the active count every CHECK iterations instead of synchronising with the GPU in each one -- and never when the
        # threshold cannot be exceeded (max - mean of a distribution over A actions is below 1 - 1/A).
        CHECK = int(getattr(p, 'check_every', 8))
        can_stop = float(p.threshold) < 1.0 - 1.0 / A
        # The tree work between two iterations' engine calls is ONE launch (efe_mcts_step): back-propagation of the previous iteration, early
        # stop, selection -- per episode in the reference's order (mcts.py:176-191).  The active count of iteration r lands in its own
        # zero-initialised word (no memset launch); the last iteration's back-propagation follows the loop.
        self.n_active_it.zero_()
        # (one episode too, since round 6: reading the active count in every iteration -- a device synchronisation -- cost 12 % of a
        # one-episode iteration, the lagged snapshot 2 %; the price is up to LAG masked iterations behind the stop.  lagged_single = False:
        # the immediate read)
        lagged = bool(getattr(p, 'lagged_check', True)) and (E > 1 or bool(getattr(p, 'lagged_single', True)))
        LAG = 2
        pending = None                                   # (iteration, sims, q0) whose back-propagation is still to run
        for repeat in range(p.repeats):
            if pending is None:
                prev, pexp = (None, None, None, 1, None, None, None), (None, None, None)
            else:
                pr, sims_t, q0_t, G_t, ps_t = pending
                prev = (p_(self.H_act[pr]), p_(self.H_len[pr]), p_(sims_t), int(p.simulation_repeats), p_(q0_t), p_(self.H_g[pr]), p_(self.H_active[pr]))
                pexp = (p_(self.n_nodes), p_(G_t), p_(ps_t))          # the previous leaf's expansion bookkeeping rides in the same launch
            self._call(lib.efe_mcts_step, prev[0], prev[1], prev[2], prev[3], prev[4], prev[5], prev[6], p_(active), p_(self.stop_at), repeat,
                       float(p.threshold), p_(self.n_active_it[repeat:]), float(p.C), 1 if p.using_prior_for_exploration else 0, self.max_depth,
                       p_(self.path_nodes), p_(self.H_act[repeat]), p_(self.H_len[repeat]), p_(self.leaf), p_(self.leaf_s), p_(self.leaf_rep),
                       pexp[0], pexp[1], pexp[2])
            pending = None
            if can_stop and E == 1 and not lagged:        # one episode: stop the loop in the iteration the episode stops in
                if int(self.n_active_it[repeat].item()) == 0:
                    break
            elif can_stop and lagged:
                # The host follows the device LAG iterations behind: a snapshot of `active` is copied to pinned memory behind every
                # step kernel, and before enqueuing iteration r the host waits for the snapshot of iteration r - LAG only (LAG
                # iterations of work stay queued: the GPU never drains, unlike a read of the current count).  From that snapshot it
                # ends the loop and re-compacts the batch (episodes stopped since then are in the batch but masked).
                slot = repeat % len(self.ev_snap)
                self.h_active[slot].copy_(active, non_blocking=True)
                self.ev_snap[slot].record(torch.cuda.current_stream(m.device))
                if repeat >= LAG:
                    ls = (repeat - LAG) % len(self.ev_snap)
                    self.ev_snap[ls].synchronize()
                    snap = self.h_active[ls].numpy()
                    n_live = int(snap.sum())
                    if n_live == 0:
                        break
                    if compact:
                        self._compact_host(snap, n_live)
            elif can_stop and repeat % CHECK == 0:
                n_live = int(self.n_active_it[repeat].item())
                if n_live == 0:
                    break
                if compact:
                    self._compact(n_live)
            st_exp = st0 + repeat * per_it
            if self.overlap:
                cur = torch.cuda.current_stream(m.device)
                self.ev_sel.record(cur)
                with torch.cuda.stream(self.sim_stream):
                    self.sim_stream.wait_event(self.ev_sel)              # leaf_s is ready; the previous back-propagation has read sims / q0
                    self._simulate(self.sim_model, active if skip else None, lambda r: st_exp + 1 + r, direct=True)
                    self.ev_sim.record(self.sim_stream)
                exp_out = self._expand(self.leaf, active, self.leaf_rep, stage=st_exp, use_mask=skip, defer=True)
                cur.wait_event(self.ev_sim)
            else:
                exp_out = self._expand(self.leaf, active, self.leaf_rep, stage=st_exp, use_mask=skip, defer=True)
                self._simulate(m, active if skip else None, lambda r: st_exp + 1 + r, direct=True)
            pending = (repeat,) + (self._sim_out if self._sim_out is not None else (self.sims, self.q0)) + exp_out
            n_iter += 1
        if pending is not None:
            pr, sims_t, q0_t, G_t, ps_t = pending
            self._call(lib.efe_mcts_expand, p_(self.n_nodes), p_(self.leaf), p_(active), p_(G_t), p_(ps_t))
            self._call(lib.efe_mcts_backprop, p_(self.path_nodes), p_(self.H_act[pr]), p_(self.H_len[pr]), p_(self.leaf), p_(active),
                       p_(sims_t), int(p.simulation_repeats), p_(q0_t), self.max_depth, p_(self.H_g[pr]), p_(self.H_active[pr]))
        # read the history back once, then plain Python lists on the host: per-element torch indexing here cost ~30 ms per 64-episode
        # decision, per-element numpy indexing 2.6 ms (during which the GPU has nothing queued); whole-array tolist() + list slicing 1.0 ms
        H_act, H_len = self.H_act[:n_iter].cpu().numpy(), self.H_len[:n_iter].cpu().numpy()
        H_g, H_active = self.H_g[:n_iter].cpu().numpy(), self.H_active[:n_iter].cpu().numpy().astype(bool)
        stop_at, N, child = self.stop_at.cpu().tolist(), self.N.cpu().numpy(), self.child.cpu().numpy()
        acts, lens, gs = H_act.transpose(1, 0, 2).tolist(), H_len.T.tolist(), H_g.T.tolist()        # [E][iteration]...
        live = H_active.T
        all_live = live.all(axis=1).tolist()
        visited = self._most_visited_paths(N, child)
        for e in range(E):
            if res[e] is not None:
                continue
            its = range(n_iter) if all_live[e] else np.flatnonzero(live[e]).tolist()
            Ae, Le, Ge = acts[e], lens[e], gs[e]
            paths = [Ae[i][:Le[i]] for i in its]
            Gs = [Ge[i] for i in its]
            reps = stop_at[e] if stop_at[e] >= 0 else p.repeats
            explored = len(paths) * p.simulation_depth * p.simulation_repeats
            res[e] = (_trim_path(visited[e], self.pi_dim), reps, explored, paths, Gs)
        return res

    @staticmethod
    def _most_visited_paths(N, child):
        """Node.action_selection's walk (mcts.py:98-110: argmax of the visit counts, first index on ties, down to the first node without
        children) for every episode at once: host arrays N [E][cap][A], child [E][cap][A] -> E lists of actions"""
        E = N.shape[0]
        ar = np.arange(E)
        node = np.zeros(E, dtype=np.int64)
        done = np.zeros(E, dtype=bool)
        out = [[] for _ in range(E)]
        for _ in range(child.shape[1]):                       # a path is shorter than the node count
            if done.all():
                break
            a = np.argmax(N[ar, node], axis=1)
            for e, (ae, d) in enumerate(zip(a.tolist(), done.tolist())):
                if not d:
                    out[e].append(ae)
            node = np.where(done, node, child[ar, node, a])
            done |= child[ar, node, 0] < 0
        return out

    def root_visit_distribution(self):
        """N / sum N at the roots, [E, pi_dim]: the policy-value that multi-GPU runs gather (mcts.py:177)"""
        n = self.N[:, 0].cpu()
        return n / n.sum(dim=1, keepdim=True)


def active_inference_mcts_batch(model, frames, params, o_shape=(64, 64, 1), episode_offset=0):
    """E planning decisions in lock-step; returns a list of E tuples shaped like active_inference_mcts's result and
    the [E, pi_dim] root visit distribution."""
    frames = torch.as_tensor(frames)
    # planner objects (device buffers) are kept per (episodes, offset, parameters) on the model
    try:
        key = (int(frames.shape[0]), int(episode_offset), tuple(sorted((k, v) for k, v in vars(params).items())))
        hash(key)
    except TypeError:
        key = None
    cache = model.__dict__.setdefault('_planners', {})
    planner = cache.get(key) if key is not None else None
    if planner is None:
        planner = BatchedMCTS(model, frames.shape[0], params, episode_offset)
        if key is not None:
            if len(cache) >= 4:
                cache.pop(next(iter(cache)))
            cache[key] = planner
    else:
        planner.reset()
    out = planner.run(frames, o_shape)
    return out, planner.root_visit_distribution()
