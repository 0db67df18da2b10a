"""Channel sharding across GPUs.  Channels are independent streams (SURVEY.md
section 8e; the reference itself builds one chain per channel, python/radio.py:88-91),
so the path shards by contiguous channel ranges with NO data-path collective:
torch.distributed is used only to line the ranks up for timing and to gather
per-rank results on rank 0."""


def shard_channels(total_channels, world_size, rank):
    """Contiguous, balanced split: returns (first_channel, count) of `rank`."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d not in [0, %d)" % (rank, world_size))
    base, extra = divmod(int(total_channels), int(world_size))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def max_over_ranks(value, device=None):
    """Max of a python float over all ranks (identity when not distributed)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(local_counts, device=None):
    """Concatenate per-rank 1-D integer arrays on every rank, in rank order
    (results are per channel: gathering is concatenation, nothing is reduced)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    a = np.ascontiguousarray(local_counts, dtype=np.int64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return a
    sizes = [None] * dist.get_world_size()
    dist.all_gather_object(sizes, int(a.size))
    m = max(sizes)
    pad = torch.zeros(m, dtype=torch.int64, device=device)
    pad[: a.size] = torch.as_tensor(a, device=device)
    outs = [torch.zeros(m, dtype=torch.int64, device=device) for _ in sizes]
    dist.all_gather(outs, pad)
    return np.concatenate([o[:s].cpu().numpy() for o, s in zip(outs, sizes)])
