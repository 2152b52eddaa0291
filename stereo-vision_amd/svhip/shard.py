"""Multi-GPU sharding of independent stereo pairs.

Pairs carry no state from one to the next (libelas/src/elas.cpp:32-170 keeps
nothing between calls), so N ranks simply take disjoint, contiguous slices of the
work list -- no data-path collective.  The only exchange is a tiny per-rank
result record gathered at the end (RCCL over xGMI on GPUs; gloo in CPU tests).
"""
import numpy as np


def shard_range(n_items, rank, world):
    """contiguous slice [lo, hi) of n_items for `rank`; sizes differ by at most one"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_records(record, dist=None, device=None, force=False):
    """all-gather one small float64 record per rank; returns [world, len(record)].
    force: run the collective even in a world of one (bench.py --force-dist: one execution of the
    RCCL path on a 1-GPU box)"""
    import torch
    rec = torch.as_tensor(np.asarray(record, np.float64), device=device)
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return rec.cpu().numpy()[None, :]
    out = [torch.zeros_like(rec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rec)
    return np.stack([o.cpu().numpy() for o in out])


def run_sharded(n_items, process_fn, dist=None, device=None):
    """each rank runs process_fn(lo, hi) -> record on its slice; records of all
    ranks come back in rank order"""
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    lo, hi = shard_range(n_items, rank, world)
    return gather_records(process_fn(lo, hi), dist, device)
