"""Single-episode MCTS planner over the engine: same public API as /root/reference/src/mcts.py
(`Node`, `MCTS_Params`, `active_inference_mcts`) so existing callers keep working.  The tree is tiny
host-side bookkeeping (pi_dim floats per node); every network evaluation goes to the HIP engine through
`model.calculate_G[_mean]` / `model.mcts_step_simulate`.
"""
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
        bonus = self.C / self.N
        return q + (self.Qpi * bonus if self.using_prior_for_exploration else bonus)

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


def active_inference_mcts(model, frame, params, o_shape=(64, 64, 1)):
    """One planning decision (mcts.py:150-195) -> (path, repeats_done, states_explored, all_paths, all_paths_G)."""
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

    qs0_mean, _ = model.model_down.encoder(torch.as_tensor(frame).reshape(1, *o_shape))
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
    """All tree statistics are [E, nodes, pi_dim] host tensors and every step is vectorised over the episodes
    (no per-episode Python in the iteration loop); node states live on the device."""

    def __init__(self, model, n_episodes, params, episode_offset=0):
        self.model, self.E, self.p = model, int(n_episodes), params
        self.pi_dim = model.pi_dim
        self.ep0 = int(episode_offset)
        self.cap = 1 + self.pi_dim * (params.repeats + 2)
        self.max_depth = params.repeats + 2
        E, A, cap = self.E, self.pi_dim, self.cap
        self.W = torch.zeros(E, cap, A)
        self.N = torch.zeros(E, cap, A)
        self.Qpi = torch.zeros(E, cap, A)
        self.child = torch.full((E, cap, A), -1, dtype=torch.long)
        self.S = torch.zeros(E, cap, model.s_dim, device=model.device)
        self.n_nodes = torch.ones(E, dtype=torch.long)
        self.ar = torch.arange(E)
        self.ar_dev = self.ar.to(model.device)
        self.arA = torch.arange(A)
        self.pi_hot = (model.pi_one_hot if A == 4 else model.pi_one_hot_3).repeat(E, 1)

    def _scores(self, e_idx, nodes):
        W, N = self.W[e_idx, nodes], self.N[e_idx, nodes]
        q = W / N
        q = q - q.min(dim=1, keepdim=True).values
        q = q / q.sum(dim=1, keepdim=True)
        bonus = self.p.C / N
        if self.p.using_prior_for_exploration:
            bonus = self.Qpi[e_idx, nodes] * bonus
        return q + bonus

    def select(self, active):
        """tree policy for every active episode at once -> (path_nodes [E,D], path_actions [E,D], path_len [E], leaf [E]);
        rows of inactive episodes are zeros"""
        E = self.E
        path_nodes = torch.zeros(E, self.max_depth, dtype=torch.long)
        path_act = torch.zeros(E, self.max_depth, dtype=torch.long)
        path_len = torch.zeros(E, dtype=torch.long)
        cur = torch.zeros(E, dtype=torch.long)
        live = active.clone()
        d = 0
        while bool(live.any()):
            e_idx = self.ar[live]
            a = torch.argmax(self._scores(e_idx, cur[e_idx]), dim=1)
            path_nodes[e_idx, d] = cur[e_idx]
            path_act[e_idx, d] = a
            path_len[e_idx] = d + 1
            nxt = self.child[e_idx, cur[e_idx], a]
            cur[e_idx] = nxt
            live = live.clone()
            live[e_idx] = self.child[e_idx, nxt, 0] >= 0          # keep walking while the reached node has children
            d += 1
        return path_nodes, path_act, path_len, cur

    def expand(self, nodes, mask):
        """ONE engine call over E x pi_dim rows; tree bookkeeping only where mask[e]"""
        m, E, A = self.model, self.E, self.pi_dim
        s = self.S[self.ar_dev, nodes.to(self.S.device)].repeat_interleave(A, dim=0)
        ro = self.ep0 * A
        if self.p.use_means:
            G, _, ps_next, _ = m.calculate_G_mean(s, self.pi_hot, row_offset=ro)
        else:
            G, _, ps_next, _, _ = m.calculate_G(s, self.pi_hot, samples=getattr(self.p, 'samples', 1), row_offset=ro)
        Gc = G.detach().to('cpu').reshape(E, A)
        e_idx = self.ar[mask]
        n = nodes[e_idx]
        self.W[e_idx, n] -= Gc[e_idx]
        self.N[e_idx, n] += 1.0
        base = self.n_nodes[e_idx]
        kids = base[:, None] + self.arA[None, :]
        self.child[e_idx, n] = kids
        dev = self.S.device
        self.S[e_idx.to(dev)[:, None], kids.to(dev)] = ps_next.reshape(E, A, -1)[e_idx.to(dev)]
        self.n_nodes[e_idx] = base + A

    def action_selection(self, e):
        visited, node = [], 0
        while True:
            a = int(torch.argmax(self.N[e, node]))
            visited.append(a)
            node = int(self.child[e, node, a])
            if self.child[e, node, 0] < 0:
                break
        return _trim_path(visited, self.pi_dim)

    def run(self, frames, o_shape=(64, 64, 1)):
        # the tree statistics are tiny host tensors: one intra-op thread (torch's default pool = every core of the
        # host, which turns each [E,4] reduction into a multi-millisecond fork/join on many-core boxes)
        prev = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            return self._run(frames, o_shape)
        finally:
            torch.set_num_threads(prev)

    def _run(self, frames, o_shape):
        m, E, p = self.model, self.E, self.p
        res = [None] * E
        explored = torch.zeros(E, dtype=torch.long)
        hist_paths, hist_G, hist_active = [], [], []          # per iteration: (path_act, path_len), G, active mask
        qs0_mean, _ = m.model_down.encoder(torch.as_tensor(frames).reshape(E, *o_shape), row_offset=self.ep0)
        self.S[:, 0] = qs0_mean
        self.Qpi[:, 0] = m.model_top.encode_s(qs0_mean)[1].to('cpu')
        active = torch.ones(E, dtype=torch.bool)
        stop_at = torch.full((E,), -1, dtype=torch.long)
        if p.use_habit:
            for e in range(E):
                if calc_threshold(self.Qpi[e, 0], axis=0) > p.threshold:
                    res[e] = ([int(torch.multinomial(self.Qpi[e, 0], 1))], 0, 0, [], [])
                    active[e] = False
        self.expand(torch.zeros(E, dtype=torch.long), active)
        for repeat in range(p.repeats):
            rootN = self.N[:, 0]
            dist = rootN / rootN.sum(dim=1, keepdim=True)
            done = active & ((dist.max(dim=1).values - dist.mean(dim=1)) > p.threshold)
            if bool(done.any()):
                for e in self.ar[done].tolist():
                    stop_at[e] = repeat
                active = active & ~done
            if not bool(active.any()):
                break
            path_nodes, path_act, path_len, leaves = self.select(active)
            self.expand(leaves, active)
            sims = torch.zeros(E, p.simulation_repeats)
            leaf_states = self.S[self.ar_dev, leaves.to(self.S.device)]
            e_idx = self.ar[active]
            for r in range(p.simulation_repeats):
                G, _, q0 = m.simulate_batch(leaf_states, p.simulation_depth, use_means=False, row_offset=self.ep0)
                sims[:, r] = G.to('cpu')
                self.Qpi[e_idx, leaves[e_idx]] = q0.to('cpu')[e_idx]
            explored[e_idx] += p.simulation_depth * p.simulation_repeats
            g = sims.mean(dim=1)
            for d in range(int(path_len.max())):
                sel = active & (path_len > d)
                ei = self.ar[sel]
                self.W[ei, path_nodes[ei, d], path_act[ei, d]] -= g[ei]
                self.N[ei, path_nodes[ei, d], path_act[ei, d]] += 1
            hist_paths.append((path_act, path_len)); hist_G.append(g); hist_active.append(active.clone())
        for e in range(E):
            if res[e] is not None:
                continue
            paths = [hist_paths[i][0][e, :int(hist_paths[i][1][e])].tolist() for i in range(len(hist_G)) if hist_active[i][e]]
            Gs = [hist_G[i][e].item() for i in range(len(hist_G)) if hist_active[i][e]]
            reps = int(stop_at[e]) if stop_at[e] >= 0 else p.repeats
            res[e] = (self.action_selection(e), reps, int(explored[e]), paths, Gs)
        return res

    def root_visit_distribution(self):
        """N / sum N at the roots, [E, pi_dim]: the policy-value that multi-GPU runs gather (mcts.py:177)"""
        n = self.N[:, 0]
        return n / n.sum(dim=1, keepdim=True)


def active_inference_mcts_batch(model, frames, params, o_shape=(64, 64, 1), episode_offset=0):
    """E planning decisions in lock-step; returns a list of E tuples shaped like active_inference_mcts's result and
    the [E, pi_dim] root visit distribution."""
    frames = torch.as_tensor(frames)
    planner = BatchedMCTS(model, frames.shape[0], params, episode_offset)
    out = planner.run(frames, o_shape)
    return out, planner.root_visit_distribution()
