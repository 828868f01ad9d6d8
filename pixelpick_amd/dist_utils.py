"""Multi-GPU plumbing of the acquisition round.  One process per GPU (torch.distributed; backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Images are independent (query.py:159 loops B=1): image i belongs to rank i mod W, nothing crosses ranks on the data path, and ONE
gather of the small per-image records (sorted picks + the statistics contribution of the image, ~100 B per image) rebuilds the
global result on every rank.  `QuerySelector.__call__` uses exactly these helpers; the train step's only exchange is the
all-reduce of the flat gradient in `trainer.FlatTrainer`.
"""
from typing import List, Tuple

import torch.distributed as dist


def rank_world(group=None) -> Tuple[int, int]:
    """(rank, world) of `group` (None = the default group); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def owner_rank(index: int, world: int) -> int:
    """Round-robin owner of item `index`."""
    return index % world


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """The items rank `rank` of `world` processes owns."""
    return list(range(rank, n_items, world))


def gather_records(records: list, world: int, group=None) -> list:
    """records: this rank's list of tuples whose first element is the global item index -> the union over ranks, sorted by
    that index (the order a single-rank loop would have produced), on EVERY rank."""
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, records, group=group)
        records = [r for part in parts for r in part]
    return sorted(records, key=lambda r: r[0])
