"""Multi-GPU sharding of a batch of independent pairs, one process per GPU.

Pairs are independent (the reference scorer is a pure function, source.cpp:462-466), so rank g scores the contiguous
sub-batch [g*N/G, (g+1)*N/G) and the only exchange step is the gather of int32 scores -- torch.distributed
all_gather_into_tensor, which is RCCL over xGMI with the "nccl" backend and plain TCP with "gloo" (CPU tests).  No input
bytes ever cross GPUs.  (The same split from ONE process driving G GPUs is swmi_score_batch_multi / swmi_sharded_* in
the C library; swmi_shard_bounds there and shard_bounds here are one rule, tests/test_abi.py.)

bench.py --gpus N runs exactly the code in this file: shard_bounds, gather_scores and GatherPipeline; the world_size-2
gloo test (tests/test_sharding_gloo.py) drives the same three with the CPU oracle standing in for the GPU scorer.
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


def gather_scores(local_scores, n_total, group=None, out=None, async_op=False):
    """All ranks receive the full int32 score vector (length n_total) in pair order.

    Equal shards use ONE all_gather_into_tensor straight into `out` (allocated when None) -- asynchronously when
    async_op is set, in which case (out, work) is returned and the caller waits on `work`.  Ragged shards are padded to
    the largest shard, gathered and compacted (synchronous; async_op then returns work = None).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        if out is not None and out.data_ptr() != local_scores.data_ptr():
            out.copy_(local_scores)
        res = out if out is not None else local_scores
        return (res, None) if async_op else res
    rank = dist.get_rank(group)
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    longest = max(hi - lo for lo, hi in sizes)
    lo, hi = sizes[rank]
    if local_scores.numel() != hi - lo:
        raise ValueError("rank %d holds %d scores, its shard has %d pairs" % (rank, local_scores.numel(), hi - lo))
    if out is None:
        out = torch.empty(n_total, dtype=local_scores.dtype, device=local_scores.device)
    if all(h - l == longest for l, h in sizes):
        work = dist.all_gather_into_tensor(out, local_scores.contiguous(), group=group, async_op=async_op)
        return (out, work) if async_op else out
    send = torch.zeros(longest, dtype=local_scores.dtype, device=local_scores.device)
    send[: hi - lo] = local_scores
    recv = torch.empty(world * longest, dtype=local_scores.dtype, device=local_scores.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    torch.cat([recv[r * longest: r * longest + (h - l)] for r, (l, h) in enumerate(sizes)], out=out)
    return (out, None) if async_op else out


class GatherPipeline:
    """K scoring steps over this rank's shard with the score gather behind them.

    mode "every": step k scores into ring slot k % depth and starts the (asynchronous) gather of that slot, which overlaps
                  the kernels of the following steps; a slot is reused only after its gather has completed.
    mode "final": the steps only score; finish() runs ONE gather of the last step's scores (BASELINE configs[3]: "per-GPU
                  sub-batch + RCCL gather").
    mode "none" : no exchange at all (single-GPU runs, and the kernel-only leg).
    `alloc(n)` returns an int32 buffer of n elements on this rank's device.
    """

    def __init__(self, n_total, rank, world, alloc, group=None, mode="every", depth=4):
        if mode not in ("every", "final", "none"):
            raise ValueError("mode must be every / final / none")
        self.n_total, self.rank, self.world, self.group, self.mode = n_total, rank, world, group, mode
        self.lo, self.hi = shard_bounds(n_total, rank, world)
        self.depth = depth if mode == "every" else 1
        self.local = [alloc(self.hi - self.lo) for _ in range(self.depth)]
        self.full = [alloc(n_total) for _ in range(self.depth)] if mode != "none" else None
        self.pending = [None] * self.depth
        self.last_slot = 0

    def step(self, k, launch, before=None, after=None):
        """launch(out) enqueues the scoring of this rank's shard into `out`; before() / after() bracket it (event records)."""
        slot = k % self.depth
        if self.pending[slot] is not None:
            self.pending[slot].wait()           # stream-side wait: the gather that read local[slot] `depth` steps ago is done
            self.pending[slot] = None
        if before is not None:
            before()
        launch(self.local[slot])
        if after is not None:
            after()
        if self.mode == "every":
            _, self.pending[slot] = gather_scores(self.local[slot], self.n_total, self.group, out=self.full[slot], async_op=True)
        self.last_slot = slot

    def drain(self):
        for s, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[s] = None

    def finish(self):
        """Complete every outstanding gather (mode "final": run the one gather now); returns the full score vector of the
        last step (mode "none": this rank's shard)."""
        self.drain()
        if self.mode == "final":
            gather_scores(self.local[self.last_slot], self.n_total, self.group, out=self.full[0])
            return self.full[0]
        if self.mode == "none":
            return self.local[self.last_slot]
        return self.full[self.last_slot]


def score_sharded(score_fn, generate_fn, n_total, group=None, mode="final", steps=1):
    """Score pairs [0, n_total) across the ranks of `group` through a GatherPipeline.

    generate_fn(first_pair, n) -> (seq1s, seq2s) for that range on this rank's device;
    score_fn(seq1s, seq2s, out) writes n int32 scores into `out`.  Returns the full score vector on every rank.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(n_total, rank, world)
    seq1s, seq2s = generate_fn(lo, hi - lo)
    device = seq1s.device
    pipe = GatherPipeline(n_total, rank, world, lambda n: torch.empty(n, dtype=torch.int32, device=device), group, mode)
    for k in range(steps):
        pipe.step(k, lambda out: score_fn(seq1s, seq2s, out))
    return pipe.finish()
