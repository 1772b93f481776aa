"""Multi-GPU sharding of planning episodes: one process per GPU (`torch.distributed`, backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests).  Episodes / root states are independent (no cross-row operation exists
anywhere in calculate_G*), so the data path has NO collective; the only exchange is the final gather of
the [episodes, pi_dim] action posteriors (<= 1 KiB per rank at BASELINE cfg-4), once per planning step.

Because every noise draw is keyed by the GLOBAL row index (csrc/philox.h), a rank that owns episodes
[start, start+count) passes row_offset = start * rows_per_episode and the gathered result is bit-identical to a
single-GPU run over all episodes.
"""
import torch


def episode_shard(n_episodes, world_size, rank):
    """Contiguous block partition -> (start, count); the first (n % world) ranks get one extra episode."""
    if not (0 <= rank < world_size):
        raise ValueError('rank out of range')
    base, extra = divmod(int(n_episodes), int(world_size))
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def gather_action_posteriors(local, n_episodes=None, group=None):
    """all-gather of per-rank [count_r, n] tensors into [n_episodes, n] on every rank (episode-major order).
    Equal shards use one all_gather_into_tensor; ragged shards are padded to the largest one."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if n_episodes is None:
        cnt = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
        dist.all_reduce(cnt, group=group)
        n_episodes = int(cnt.item())
    counts = [episode_shard(n_episodes, world, r)[1] for r in range(world)]
    if counts[rank] != local.shape[0]:
        raise ValueError(f'rank {rank} holds {local.shape[0]} episodes, expected {counts[rank]}')
    n = local.shape[1]
    if len(set(counts)) == 1:
        out = torch.empty(n_episodes, n, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    mx = max(counts)
    padded = torch.zeros(mx, n, dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    buf = torch.empty(world * mx, n, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * mx:r * mx + counts[r]] for r in range(world)], dim=0)
