"""The N > 1 path on CPU: world_size-2 (and -8) gloo processes run swmi/sharding.py -- shard_bounds, gather_scores and the
GatherPipeline, the very code bench.py --gpus N runs -- with the CPU oracle standing in for the GPU scorer (this test
checks host logic only): gather after every step (asynchronous, ring of buffers) and one final gather, equal and
ragged shards."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT, Oracle, match_matrix


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_total, out_dir, mode, steps):
    import sys
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from swmi import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = Oracle()
    sm = match_matrix(10, -30)

    def generate(first, n):
        a, b = oracle.generate(n, 10000, first)
        return torch.from_numpy(a), torch.from_numpy(b)

    def score(a, b, out):
        out.copy_(torch.from_numpy(oracle.batch(a.numpy(), b.numpy(), sm, 15)))

    full = sharding.score_sharded(score, generate, n_total, mode=mode, steps=steps)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total,mode,steps", [(4096, "final", 1), (4097, "final", 1), (4096, "every", 6), (4097, "every", 3)])
def test_two_rank_shard_and_gather(tmp_path, oracle, n_total, mode, steps):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path), mode, steps), nprocs=world, join=True)
    a, b = oracle.generate(n_total, 10000, 0)
    want = oracle.batch(a, b, match_matrix(10, -30), 15)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got, want), "rank %d" % r


@pytest.mark.parametrize("n_total,mode,steps", [(4099, "every", 3), (5, "final", 1)])
def test_eight_rank_shard_and_gather(tmp_path, oracle, n_total, mode, steps):
    """The size the driver's scaling run ends at: eight ranks, ragged shards (4099 = 3 x 513 + 5 x 512), and fewer pairs than
    ranks (three ranks with an empty shard take part in the gathers all the same)."""
    world = 8
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path), mode, steps), nprocs=world, join=True)
    a, b = oracle.generate(n_total, 10000, 0)
    want = oracle.batch(a, b, match_matrix(10, -30), 15)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got, want), "rank %d" % r


def test_shard_bounds_partition():
    from swmi import sharding
    for n in (0, 1, 7, 8, 1000003, 1 << 20):
        for world in (1, 2, 3, 4, 8):
            bounds = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == n
            for (l0, h0), (l1, h1) in zip(bounds, bounds[1:]):
                assert h0 == l1
            sizes = [h - l for l, h in bounds]
            assert max(sizes) - min(sizes) <= 1
