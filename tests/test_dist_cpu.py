"""World-size-2 gloo tests (CPU) of the acquisition round's N>1 plumbing (pixelpick_amd/dist_utils.py): ownership, and the ONE
gather of per-image records that rebuilds the single-rank order on every rank."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from pixelpick_amd import dist_utils as du


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    assert du.rank_world() == (0, 1)                       # not initialised yet: single rank
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert du.rank_world() == (rank, world)
        n_items, k = 7, 5
        mine = [i for i in range(n_items) if du.owner_rank(i, world) == rank]
        assert mine == du.shard_indices(n_items, rank, world)
        # the record QuerySelector ships per image: (index, path, h, w, sorted picks, statistics contribution)
        local = [(i, f"img{i}.png", 4, 8, np.arange(i * 100, i * 100 + k, dtype=np.int64), ([i] * k, [0.5 * i] * k, 1, 2.0)) for i in mine]
        full = du.gather_records(local, world)
        ok = [r[0] for r in full] == list(range(n_items))
        ok = ok and all(np.array_equal(r[4], np.arange(r[0] * 100, r[0] * 100 + k)) and r[1] == f"img{r[0]}.png" for r in full)
        ok = ok and all(r[5][0] == [r[0]] * k for r in full)
        q.put((rank, ok, mine))
    finally:
        dist.destroy_process_group()


def test_ownership_and_record_gather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][2] == [0, 2, 4, 6] and res[1][2] == [1, 3, 5]
    assert all(r[1] for r in res)


def _rate_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank reports its own units and its own clock: the job's figure is the SUM of the units over the SLOWEST rank's time
        q.put((rank,) + du.whole_job_rate(units_local=120.0 * (rank + 1), elapsed_local=1.0 + 0.25 * rank))
    finally:
        dist.destroy_process_group()


def test_whole_job_rate_aggregates_all_eight_ranks():
    """bench.py's `value` contract at N = 8 (the driver's SCALE run): total units of all ranks / max-over-ranks time, the same
    number on every rank; a single process is its own total."""
    assert du.whole_job_rate(30.0, 2.0) == (15.0, 30.0, 2.0)
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rate_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    units = 120.0 * sum(range(1, world + 1))
    for r in res:
        assert r[1:] == (units / 2.75, units, 2.75)


def test_shard_indices_cover_everything_once():
    for n in (0, 1, 7, 2975):
        for w in (1, 2, 4, 8):
            seen = sorted(i for r in range(w) for i in du.shard_indices(n, r, w))
            assert seen == list(range(n))
            assert all(du.owner_rank(i, w) == r for r in range(w) for i in du.shard_indices(n, r, w))


def test_single_rank_gather_is_a_sort():
    assert du.gather_records([(2, "c"), (0, "a"), (1, "b")], 1) == [(0, "a"), (1, "b"), (2, "c")]


def test_sharded_batch_sampler_partitions_the_single_process_order():
    """dist_utils.ShardedBatchSampler: the union of the ranks' batches IS the global batch sequence (batch i -> rank i mod W),
    no index is loaded twice, training shards have equal step counts, a new epoch is a new shared permutation, and without
    shuffling the order is the reference's sequential loader."""
    from pixelpick_amd.dist_utils import ShardedBatchSampler as S
    for n, bs, w in ((10, 4, 2), (2975, 4, 8), (7, 1, 2), (5, 2, 4)):
        for shuffle in (False, True):
            for equal in (False, True):
                ranks = [S(n, bs, r, w, shuffle=shuffle, equal_steps=equal, seed=11) for r in range(w)]
                for s in ranks:
                    s.set_epoch(3)
                glob = ranks[0].global_batches()
                assert all(s.global_batches() == glob for s in ranks)
                per = [list(s) for s in ranks]
                assert all(len(p) == len(s) for p, s in zip(per, ranks))
                inter = [b for i in range(max(len(p) for p in per)) for p in per if i < len(p) for b in [p[i]]]
                assert inter == glob
                flat = [i for b in glob for i in b]
                assert len(flat) == len(set(flat))
                if equal:
                    assert len({len(p) for p in per}) == 1 and len(glob) % w == 0
                else:
                    assert sorted(flat) == list(range(n))
                if not shuffle:
                    assert flat == list(range(len(flat)))
    a = S(100, 4, 0, 2, shuffle=True, seed=1)
    e0 = list(a)
    a.set_epoch(1)
    assert list(a) != e0


def test_shard_dataloader_loads_only_own_items():
    import torch
    from pixelpick_amd.dist_utils import shard_dataloader, dataset_image_sizes

    class DS(torch.utils.data.Dataset):
        image_sizes = [(4, 6)] * 9

        def __init__(self):
            self.loaded = []

        def __len__(self):
            return 9

        def __getitem__(self, i):
            self.loaded.append(i)
            return {"x": torch.full((1,), float(i)), "p_img": f"{i}"}

    ds = DS()
    dl = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=True, drop_last=True)
    sh = shard_dataloader(dl, 1, 2, equal_steps=True, seed=5)
    got = [b["x"].tolist() for b in sh]
    assert len(got) == 2 and len(ds.loaded) == 4                     # 4 global batches of 2 -> 2 per rank; 4 items touched
    assert shard_dataloader([1, 2, 3], 0, 2, True) is None          # not a DataLoader: caller falls back
    assert dataset_image_sizes(ds) == [(4, 6)] * 9 and dataset_image_sizes(object()) is None


def test_rccl_channel_count_is_read_from_the_init_log(tmp_path):
    """bench.py --gpus N reports the channels the communicator opened (one resident block per channel) beside pp_get_comm_cu_reserve():
    parsed from the NCCL_DEBUG=INFO INIT lines of the run."""
    new = ("box:123:456 [0] NCCL INFO comm 0x55 rank 0 nranks 8 cudaDev 0 busId c000 - Init COMPLETE\n"
           "box:123:456 [0] NCCL INFO 28 coll channels, 0 collnet channels, 0 nvls channels, 32 p2p channels, 4 p2p channels per peer\n")
    old = "box:1:2 [0] NCCL INFO Channel 00/16 :    0   1   2   3\nbox:1:2 [0] NCCL INFO Channel 15/16 :    0   3   2   1\n"
    assert du.rccl_channels_from_log(new) == 28 and du.rccl_channels_from_log(old) == 16 and du.rccl_channels_from_log("nothing\nhere\n") is None
    p = tmp_path / "rccl.log"
    p.write_text(new)
    assert du.rccl_channels_from_log(str(p)) == 28


def test_comm_cu_reserve_default_and_restore(monkeypatch):
    """trainer.py: the reserve defaults to one CU per RCCL channel (NCCL_MAX_NCHANNELS, else 32; 0 for gloo), and the last collective
    trainer to close puts the library's process-wide setting back to what it was before the first one changed it."""
    from pixelpick_amd import _lib, trainer as T
    from pixelpick_amd import engine as E
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    assert T.default_comm_cu_reserve("nccl") == 32 and T.default_comm_cu_reserve("gloo") == 0
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "48")
    assert T.default_comm_cu_reserve("nccl") == 48
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "junk")
    assert T.default_comm_cu_reserve("nccl") == 32
    L = _lib.lib()
    L.pp_set_comm_cu_reserve(5)
    try:
        def fake():
            t = T.FlatTrainer.__new__(T.FlatTrainer)
            t._plan = t._plan_pool = t._gx = t._gy = None
            t._seed_dev = object()
            if T._RESERVE["users"] == 0:
                T._RESERVE["before"] = L.pp_get_comm_cu_reserve()
            T._RESERVE["users"] += 1
            t._holds_reserve = True
            L.pp_set_comm_cu_reserve(32)
            return t
        a, b = fake(), fake()
        assert L.pp_get_comm_cu_reserve() == 32
        a.close()
        assert L.pp_get_comm_cu_reserve() == 32            # b still steps under a resident communicator
        b.close()
        b.close()                                          # idempotent
        assert L.pp_get_comm_cu_reserve() == 5 and T._RESERVE == {"users": 0, "before": None}
    finally:
        L.pp_set_comm_cu_reserve(0)
