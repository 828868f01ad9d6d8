"""Multi-GPU plumbing of the acquisition round.  One process per GPU (torch.distributed; backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Images are independent (query.py:159 loops B=1): image i belongs to rank i mod W, nothing crosses ranks on the data path, and ONE
gather of the small per-image records (sorted picks + the statistics contribution of the image, ~100 B per image) rebuilds the
global result on every rank.  `QuerySelector.__call__` uses exactly these helpers; the train step's only exchange is the
all-reduce of the flat gradient in `trainer.FlatTrainer`.
"""
import os
from typing import List, Tuple

import torch.distributed as dist


def rank_world(group=None) -> Tuple[int, int]:
    """(rank, world) of `group` (None = the default group); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def whole_job_rate(units_local: float, elapsed_local: float, device=None, group=None) -> Tuple[float, float, float]:
    """What bench.py reports as `value`: the units ALL ranks processed in the timed region / the slowest rank's time.
    -> (rate, units over all ranks, max elapsed).  One all-reduce each (SUM, MAX) when torch.distributed runs, nothing otherwise."""
    import torch
    units, el = float(units_local), float(elapsed_local)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dev = device if device is not None else "cpu"
        u = torch.tensor([units], dtype=torch.float64, device=dev)
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(u, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        units, el = float(u.item()), float(t.item())
    return units / el, units, el


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


# ------------------------------------------------------------------------------------------------- host side of the loaders
# SURVEY.md 8(e) asks for DistributedSampler-style disjoint shards.  Skipping other ranks' batches AFTER the loader has read,
# augmented and collated them costs W times the host I/O of the single-process reference and makes the shards depend on every
# rank holding identical python / numpy / torch RNG states.  The helpers below shard at the INDEX level: each rank's loader only
# ever touches its own items, and the order comes from an explicit seed shared by construction.
import torch
from torch.utils.data import DataLoader, RandomSampler, Sampler, SequentialSampler


class ShardedBatchSampler(Sampler):
    """Batches of dataset indices for rank `rank` of `world`.

    The GLOBAL batch sequence is the one a single process would draw - a permutation of range(n) from
    `torch.Generator().manual_seed(seed + epoch)` when `shuffle`, else 0..n-1, cut into batches of `batch_size` - and global
    batch i belongs to rank i mod world.  `equal_steps` (training): the ragged tail of global batches is dropped so that every
    rank runs the same number of steps (a collective per step); validation / acquisition keep every item.
    Only indices are materialised here; no rank loads another rank's items."""

    def __init__(self, n_items: int, batch_size: int, rank: int, world: int, shuffle: bool = False, drop_last: bool = False,
                 equal_steps: bool = False, seed: int = 0):
        self.n, self.bs, self.rank, self.world = int(n_items), int(batch_size), int(rank), int(world)
        self.shuffle, self.drop_last, self.equal_steps, self.seed = shuffle, drop_last, equal_steps, int(seed)
        self.epoch = 0

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def global_batches(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        batches = [order[i:i + self.bs] for i in range(0, self.n, self.bs)]
        if self.drop_last and batches and len(batches[-1]) < self.bs:
            batches.pop()
        if self.equal_steps:
            batches = batches[:len(batches) // self.world * self.world]
        return batches

    def global_index_of_local(self):
        """Positions (in the global batch sequence) of this rank's batches."""
        return list(range(self.rank, len(self.global_batches()), self.world))

    def __iter__(self):
        gb = self.global_batches()
        return iter(gb[self.rank::self.world])

    def __len__(self):
        return len(range(self.rank, len(self.global_batches()), self.world))


def shard_dataloader(dl, rank: int, world: int, equal_steps: bool, seed: int = 0):
    """A DataLoader over the SAME dataset / collate / workers as `dl` whose batch sampler is this rank's ShardedBatchSampler.
    Returns None when `dl` is not a torch DataLoader with an integer batch size (the caller then falls back to enumerating
    the whole loader and skipping, with the identical-RNG requirement that implies)."""
    if not isinstance(dl, DataLoader) or dl.batch_size is None:
        return None
    if type(dl.sampler) not in (RandomSampler, SequentialSampler) or getattr(dl.sampler, "replacement", False) \
            or getattr(dl.sampler, "_num_samples", None) is not None:
        return None                  # subset / weighted / user samplers: their draws cannot be re-stated here - enumerate and skip
    shuffle = isinstance(dl.sampler, RandomSampler)
    bs = ShardedBatchSampler(len(dl.dataset), dl.batch_size, rank, world, shuffle=shuffle, drop_last=dl.drop_last,
                             equal_steps=equal_steps, seed=seed)
    kw = dict(batch_sampler=bs, collate_fn=dl.collate_fn, num_workers=dl.num_workers, pin_memory=dl.pin_memory,
              worker_init_fn=_rank_worker_init(dl.worker_init_fn, rank, seed), timeout=dl.timeout, generator=dl.generator)
    if dl.num_workers > 0:
        kw.update(persistent_workers=dl.persistent_workers, prefetch_factor=dl.prefetch_factor,
                  multiprocessing_context=dl.multiprocessing_context)
    return DataLoader(dl.dataset, **kw)


class _rank_worker_init:
    """worker_init_fn of a rank's sharded loader.  The acquisition round needs identical host RNG streams on every rank, so all ranks
    seed alike - and their loader workers (base seed = a draw from that stream) would then apply the SAME augmentation sequence to their
    different shards.  Each worker's python / numpy / torch generators are therefore re-seeded from (seed, rank, worker id) before the
    caller's own worker_init_fn runs.  Picklable (spawned workers)."""

    def __init__(self, inner, rank: int, seed: int):
        self.inner, self.rank, self.seed = inner, int(rank), int(seed)

    def __call__(self, worker_id: int):
        import random

        import numpy as np
        s = (self.seed * 1000003 + self.rank * 9973 + worker_id * 101 + torch.initial_seed()) % (1 << 31)
        random.seed(s)
        np.random.seed(s)
        torch.manual_seed(s)
        if self.inner is not None:
            self.inner(worker_id)


def augment_seed(seed: int, rank: int) -> int:
    """Seed of a rank's training-time augmentation draws (DeviceAugmenter generator): separate from the streams the
    acquisition draws use, different per rank."""
    return (int(seed) * 1000003 + 7919 * (int(rank) + 1)) % (1 << 31)


def dataset_image_sizes(dataset):
    """[(h, w)] of every item WITHOUT loading it, if the dataset says so (`image_sizes` attribute or `image_size(i)` method),
    else None.  A sharded acquisition round needs it to advance the host RNG streams past the images of other ranks."""
    sizes = getattr(dataset, "image_sizes", None)
    if sizes is not None:
        return [tuple(int(v) for v in s) for s in sizes]
    fn = getattr(dataset, "image_size", None)
    if callable(fn):
        return [tuple(int(v) for v in fn(i)) for i in range(len(dataset))]
    return None



def rccl_channels_from_log(path_or_text: str):
    """The number of collective channels an RCCL / NCCL communicator opened, from its NCCL_DEBUG=INFO (INIT) log: the line
    `... NCCL INFO <n> coll channels, <m> collnet channels, ... p2p channels ...` of communicator setup (older builds:
    `Channel <i>/<n> :` ring lines).  Each channel keeps one block resident on a CU for the length of a collective - the figure
    pp_set_comm_cu_reserve / PIXELPICK_COMM_CU_RESERVE has to cover.  None when the log holds neither."""
    import re
    text = path_or_text
    if "\n" not in path_or_text and os.path.exists(path_or_text):
        with open(path_or_text, errors="replace") as f:
            text = f.read()
    m = re.findall(r"(\d+) coll channels", text)
    if m:
        return max(int(v) for v in m)
    m = re.findall(r"Channel (\d+)/(\d+) ?:", text)
    if m:
        return max(int(n) for _, n in m)
    return None
