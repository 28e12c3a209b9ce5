"""Multi-GPU sharding of a batch of independent pairs.

One process per GPU.  Pairs are independent (the reference scorer is a pure function, source.cpp:462-466),
so rank g scores the contiguous sub-batch [g*N/G, (g+1)*N/G) and the only exchange step is the final gather
of int32 scores -- torch.distributed all_gather_into_tensor, which is RCCL over xGMI with the "nccl" backend
and plain TCP with "gloo" (CPU tests).  No input bytes ever cross GPUs.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_total, rank, world):
    """Contiguous shard [lo, hi) of rank `rank`; shards differ in size by at most one pair."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def gather_scores(local_scores, n_total, group=None):
    """All ranks receive the full int32 score vector (length n_total) in pair order.

    Equal shards use one all_gather_into_tensor; ragged shards are padded to the largest shard first.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_scores
    rank = dist.get_rank(group)
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    longest = max(hi - lo for lo, hi in sizes)
    lo, hi = sizes[rank]
    if local_scores.numel() != hi - lo:
        raise ValueError("rank %d holds %d scores, its shard has %d pairs" % (rank, local_scores.numel(), hi - lo))
    send = local_scores
    if hi - lo != longest:
        send = torch.zeros(longest, dtype=local_scores.dtype, device=local_scores.device)
        send[: hi - lo] = local_scores
    recv = torch.empty(world * longest, dtype=local_scores.dtype, device=local_scores.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    if all(h - l == longest for l, h in sizes):
        return recv
    return torch.cat([recv[r * longest: r * longest + (h - l)] for r, (l, h) in enumerate(sizes)])


def score_sharded(score_fn, generate_fn, n_total, group=None):
    """Score pairs [0, n_total) across the ranks of `group`.

    generate_fn(first_pair, n) -> (seq1s, seq2s) for that range on this rank's device;
    score_fn(seq1s, seq2s) -> int32 tensor of n scores on this rank's device.
    Returns the full score vector on every rank.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(n_total, rank, world)
    seq1s, seq2s = generate_fn(lo, hi - lo)
    local = score_fn(seq1s, seq2s)
    return gather_scores(local, n_total, group)
