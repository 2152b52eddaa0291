"""N>1 path on CPU: world_size-2 gloo processes shard the pair list exactly like
bench.py does on GPUs (svhip/shard.py) and gather their per-rank records."""
import os
import sys

import numpy as np
import pytest
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


def test_bench_self_spawns_ranks():
    """`python bench.py --gpus 2` from a plain shell (no WORLD_SIZE) starts two ranks through
    torch.distributed.run; without a GPU each of them stops with the no-CPU-path message"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""      # the check must not depend on the box it runs on
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0", "--dist-backend", "gloo"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0
    # (the launcher may stop the second rank as soon as the first one has failed)
    assert "bench.py needs a GPU" in r.stderr and "local_rank: " in r.stderr, r.stderr[:3000]


def _elas_worker(rank, world, port, n_items, q):
    """a rank of the real thing: its contiguous shard of the pair list through the HIP path"""
    import torch.distributed as dist
    sys.path.insert(0, H.PKG)
    import svhip as S
    from svhip import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S.lib().svh_set_device(rank % S.device_count())
    pairs = [H.synth_pair(320, 200, 900 + i, dmax=40) for i in range(n_items)]

    def process(lo, hi):
        if hi <= lo:
            return [0.0, 0.0, 0.0, float(lo)]
        st, D1, D2 = S.Elas(H.robotics()).process_batch(np.stack([p[0] for p in pairs[lo:hi]]),
                                                        np.stack([p[1] for p in pairs[lo:hi]]))
        assert all(s == 0 for s in st)
        return [float(hi - lo), float(D1.astype(np.float64).sum()), float((D2 >= 0).sum()), float(lo)]

    rec = shard.run_sharded(n_items, process, dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, rec.tolist()))


@pytest.mark.gpu
def test_two_ranks_run_real_elas_shards():
    """two ranks (gloo, sharing the GPUs that are there) run ELAS on disjoint shards;
    the gathered records add up to what one process computes for the whole list"""
    sys.path.insert(0, H.PKG)
    import svhip as S
    world, n_items = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_elas_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pairs = [H.synth_pair(320, 200, 900 + i, dmax=40) for i in range(n_items)]
    st, D1, D2 = S.Elas(H.robotics()).process_batch(np.stack([p[0] for p in pairs]),
                                                    np.stack([p[1] for p in pairs]))
    assert all(s == 0 for s in st)
    rec = np.array(got[0])
    assert got[0] == got[1] and rec.shape == (2, 4)
    assert rec[:, 0].sum() == n_items and rec[0, 3] == 0 and rec[1, 3] == rec[0, 0]
    per = [D1[int(rec[r, 3]):int(rec[r, 3] + rec[r, 0])].astype(np.float64).sum() for r in range(world)]
    assert rec[0, 1] == per[0] and rec[1, 1] == per[1]          # bit-identical maps shard by shard
    assert rec[:, 2].sum() == (D2 >= 0).sum()


@pytest.mark.gpu
def test_bench_two_ranks_on_this_box():
    """the bench's own N>1 path end to end: self-spawn, gloo (ranks may share a GPU), record gather"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--batch", "64", "--spinup", "0", "--dist-backend", "gloo",
                        "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and len(d["ranks"]) == 2
    assert sum(x["pairs"] for x in d["ranks"]) == 2 * 2 * 64
    assert d["value"] > 0 and d["scaling"] == "weak"


def test_placement_check_and_bus_id_packing():
    """8-GPU readiness (round 6): PCI addresses travel in the per-rank records as one number; rank 0 refuses a run in
    which two ranks landed on one device although there were enough devices, and reports sharing otherwise"""
    from svhip import shard
    for s in ("0000:c5:00.0", "0001:03:1f.7", "ffff:ff:00.1"):
        assert shard.unpack_bus_id(shard.pack_bus_id(s)) == s
        assert float(shard.pack_bus_id(s)) == shard.pack_bus_id(s)            # exact in the float64 records
    assert shard.pack_bus_id("") == 0 and shard.pack_bus_id("garbage") == 0 and shard.unpack_bus_id(0) == ""
    eight = [(r, "0000:%02x:00.0" % (0x05 + 16 * r)) for r in range(8)]
    ok = shard.check_placement(eight, 8)
    assert ok["ok"] and ok["one_rank_per_device"] and ok["ranks_seen"] == 8
    twice = list(eight)
    twice[5] = (3, eight[3][1])
    bad = shard.check_placement(twice, 8)
    assert not bad["ok"] and "same device or PCI address" in bad["why"]
    same_bus = list(eight)
    same_bus[7] = (7, eight[0][1])                                           # distinct indices, one address
    assert not shard.check_placement(same_bus, 8)["ok"]
    sharing = shard.check_placement([(0, "0000:05:00.0")] * 8, 1)             # the 1-GPU dry runs
    assert sharing["ok"] and not sharing["one_rank_per_device"]
