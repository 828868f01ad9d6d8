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


def test_shard_indices_cover_everything_once():
    for n in (0, 1, 7, 2975):
        for w in (1, 2, 4, 8):
            seen = sorted(i for r in range(w) for i in du.shard_indices(n, r, w))
            assert seen == list(range(n))
            assert all(du.owner_rank(i, w) == r for r in range(w) for i in du.shard_indices(n, r, w))


def test_single_rank_gather_is_a_sort():
    assert du.gather_records([(2, "c"), (0, "a"), (1, "b")], 1) == [(0, "a"), (1, "b"), (2, "c")]
