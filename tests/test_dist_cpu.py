"""World-size-2 gloo tests (CPU) of the N>1 plumbing: image sharding + gather of picks, gradient averaging."""
import os
import socket

import pytest
import torch
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
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_items, k = 7, 5
        mine = du.shard_indices(n_items, rank, world)
        # every "image" i yields the picks [i*100 .. i*100+k)
        local = torch.tensor([[i * 100 + j for j in range(k)] for i in mine], dtype=torch.int32)
        full = du.gather_sharded_rows(local, n_items, rank, world)
        expect = torch.tensor([[i * 100 + j for j in range(k)] for i in range(n_items)], dtype=torch.int32)
        ok_gather = torch.equal(full, expect)
        g = torch.full((1000,), float(rank + 1))
        du.all_reduce_mean_(g, world)
        ok_mean = torch.allclose(g, torch.full((1000,), (1 + 2) / 2.0))
        q.put((rank, ok_gather, ok_mean, mine))
    finally:
        dist.destroy_process_group()


def test_sharding_gather_and_gradient_mean_world2():
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
    assert res[0][3] == [0, 2, 4, 6] and res[1][3] == [1, 3, 5]
    assert all(r[1] and r[2] for r in res)


def test_shard_indices_cover_everything_once():
    for n in (0, 1, 7, 2975):
        for w in (1, 2, 4, 8):
            seen = sorted(i for r in range(w) for i in du.shard_indices(n, r, w))
            assert seen == list(range(n))
