"""ORACLE -- test infrastructure only (never imported by the product path).

CPU restatement of the reference planner `/root/reference/src/mcts.py` (Node :11-128, calc_threshold / normalization
:130-135, MCTS_Params :137-148, active_inference_mcts :150-195) on top of `oracle/efe_oracle.py::OracleModel`.
Tree statistics are torch CPU tensors evaluated with the reference's own expressions, so fp32 rounding, NaN
propagation (an unvisited edge gives Q = 0/0) and argmax tie-breaking are torch's, as in the reference.

Noise addressing (what the fixtures of oracle/make_golden.py and the HIP planners use): one `stage` per engine-level
call -- root encode, every expansion, every simulation -- starting at `stage0`; episode e draws at global rows
e (root encode, simulation steps), 4e + a (expansion rows) and e * depth + t (trajectory rows).

Pinned against the shimmed reference planner by tests/test_oracle_golden.py (fixtures mcts_means, mcts_samples,
mcts_prior, mcts_batch_s10)."""
import torch

from . import philox as PX


class Params:
    """mcts.py:137-148 defaults; `samples` = MC samples per expansion (Node.expand's argument, mcts.py:64)"""

    def __init__(self, **kw):
        self.C = 1.0
        self.threshold = 0.5
        self.repeats = 300
        self.simulation_repeats = 1
        self.simulation_depth = 3
        self.use_habit = False
        self.use_means = True
        self.using_prior_for_exploration = False
        self.samples = 1
        for k, v in kw.items():
            setattr(self, k, v)


class Node:
    """mcts.py:11-34"""

    def __init__(self, s, C, pi_dim, using_prior):
        self.s = torch.stack([s] * pi_dim)
        self.W = torch.zeros(pi_dim)
        self.N = torch.zeros(pi_dim)
        self.Qpi = torch.zeros(pi_dim)
        self.children = [None] * pi_dim
        self.C, self.pi_dim, self.using_prior = C, pi_dim, using_prior
        self.in_progress = -1

    def probs_for_selection(self):
        """mcts.py:39-47"""
        q = self.W / self.N
        q = q - q.min()
        q = q / q.sum()
        if self.using_prior:
            return q + self.C * self.Qpi * 1.0 / self.N
        return q + self.C * 1.0 / self.N

    def select(self):
        """mcts.py:49-62 (deterministic)"""
        path, actions = [], []
        node = self
        while True:
            node.in_progress = int(torch.argmax(node.probs_for_selection()))
            actions.append(node.in_progress)
            node = node.children[node.in_progress]
            path.append(node)
            if None in node.children:
                return path, actions

    def action_selection(self):
        """mcts.py:98-128 (deterministic)"""
        path = [int(torch.argmax(self.N))]
        node = self.children[path[-1]]
        while None not in node.children:
            path.append(int(torch.argmax(node.N)))
            node = node.children[path[-1]]
        opposite = {4: {(0, 1), (1, 0), (2, 3), (3, 2)}, 3: {(1, 2), (2, 1)}}[self.pi_dim]
        trimmed, i = [], 0
        while i < len(path) - 1:
            if (path[i], path[i + 1]) in opposite:
                i += 2
            else:
                trimmed.append(path[i])
                i += 1
        return trimmed


def calc_threshold(P):
    """mcts.py:130-131"""
    return torch.max(P, dim=0).values - torch.mean(P, dim=0)


def plan(orc, frame, params, stage0, episode=0):
    """active_inference_mcts (mcts.py:150-195) for one episode.  frame: anything reshapeable to [1, C, R, R] of the oracle's geometry.
    -> (final_path, repeats_done, states_explored, all_paths, all_paths_G, root_N)"""
    A = orc.pi_dim
    stage = [int(stage0)]

    def take():
        stage[0] += 1
        return stage[0] - 1

    def expand(node):
        """mcts.py:64-86"""
        pi_hot = torch.eye(A)
        if params.use_means:
            G, _, ps_next, _ = orc.calculate_G_mean(node.s, pi_hot, take(), A * episode)
        else:
            G, _, ps_next, _, _ = orc.calculate_G(node.s, pi_hot, params.samples, take(), A * episode)
        node.W -= G
        node.N += 1.0
        for a in range(A):
            node.children[a] = Node(ps_next[a], params.C, A, params.using_prior_for_exploration)

    with torch.no_grad():
        qs0_mean, _ = orc.encoder(torch.as_tensor(frame).reshape(1, orc.channels, orc.resolution, orc.resolution), PX.PASS_ROOT, 0, take(), episode)
        root = Node(qs0_mean[0], params.C, A, params.using_prior_for_exploration)
        root.Qpi = orc.encode_s(qs0_mean)[1][0]
        all_paths, all_G, explored = [], [], 0
        if params.use_habit and calc_threshold(root.Qpi) > params.threshold:
            return [int(torch.multinomial(root.Qpi, 1))], 0, explored, all_paths, all_G, root.N.clone()
        expand(root)
        for repeat in range(params.repeats):
            if calc_threshold(root.N / root.N.sum(dim=0)) > params.threshold:
                return root.action_selection(), repeat, explored, all_paths, all_G, root.N.clone()
            path, actions = root.select()
            leaf = path[-1]
            expand(leaf)
            sims = torch.zeros(params.simulation_repeats)
            for k in range(params.simulation_repeats):
                explored += params.simulation_depth
                g, _, qpi = orc.mcts_step_simulate(leaf.s[0], params.simulation_depth, False, take(), episode=episode)
                sims[k] = g
                leaf.Qpi = qpi
            g = sims.mean()
            for node in [root] + path[:-1]:            # backpropagate, mcts.py:88-96
                node.W[node.in_progress] -= g
                node.N[node.in_progress] += 1
                node.in_progress = -2
            all_paths.append(actions)
            all_G.append(g.item())
        return root.action_selection(), params.repeats, explored, all_paths, all_G, root.N.clone()
