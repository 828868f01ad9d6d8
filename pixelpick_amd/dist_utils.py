"""Multi-GPU plumbing for the two shardable parts of the path.  One process per GPU
(torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Acquisition: images are independent (query.py:159 loops B=1) -> image i goes to rank i mod W, no collective
on the data path; one all_gather of the [n_local, k] int32 picks rebuilds the global order on every rank.
Training: the only exchange is ONE all-reduce of the flat gradient per step (trainer.FlatTrainer).
"""
from typing import List, Optional

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin shard: the items rank `rank` of `world` processes."""
    return list(range(rank, n_items, world))


def gather_sharded_rows(local_rows: torch.Tensor, n_items: int, rank: int, world: int, group=None) -> torch.Tensor:
    """local_rows [n_local, k] (row j belongs to global item rank + j*world) -> [n_items, k] on every rank."""
    if world == 1:
        return local_rows
    k = local_rows.shape[1]
    n_max = (n_items + world - 1) // world
    pad = torch.zeros((n_max, k), dtype=local_rows.dtype, device=local_rows.device)
    pad[: local_rows.shape[0]] = local_rows
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    out = torch.empty((n_items, k), dtype=local_rows.dtype, device=local_rows.device)
    for r in range(world):
        idx = shard_indices(n_items, r, world)
        out[idx] = bufs[r][: len(idx)]
    return out


def all_reduce_mean_(flat: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """In-place mean over ranks of a flat gradient buffer (the reference semantics of a W-times larger batch
    when every image carries the same number of labelled pixels)."""
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
    return flat
