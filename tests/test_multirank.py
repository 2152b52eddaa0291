"""N>1 path on CPU: world_size-2 gloo processes shard the pair list exactly like
bench.py does on GPUs (svhip/shard.py) and gather their per-rank records."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

import helpers as H


def _worker(rank, world, port, n_items, q):
    import torch.distributed as dist
    sys.path.insert(0, H.PKG)
    from svhip import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    items = np.arange(n_items, dtype=np.float64) * 3.0 + 1.0

    def process(lo, hi):            # stand-in for "run ELAS on pairs lo..hi"
        return [hi - lo, items[lo:hi].sum(), lo]

    rec = shard.run_sharded(n_items, process, dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, rec.tolist()))


def test_two_ranks_cover_all_pairs_once():
    world, n_items = 2, 37
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    items = np.arange(n_items, dtype=np.float64) * 3.0 + 1.0
    for r in range(world):
        rec = np.array(got[r])
        assert rec.shape == (world, 3)
        assert rec[:, 0].sum() == n_items                      # every pair exactly once
        assert abs(rec[:, 1].sum() - items.sum()) < 1e-9       # same result as one process
        assert rec[0, 2] == 0 and rec[1, 2] == rec[0, 0]       # contiguous, rank order
    assert got[0] == got[1]                                    # every rank sees the same gather


def test_shard_ranges():
    sys.path.insert(0, H.PKG)
    from svhip import shard
    for n in (0, 1, 7, 64, 430):
        for world in (1, 2, 4, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
