"""World-size-2 test of the data-parallel train step ON THE GPU: two processes share cuda:0 and exchange the flat
gradient through gloo (RCCL wants one device per rank; gloo accepts device tensors), so the whole N>1 path of
FlatTrainer - shard-local forward/backward, ONE all-reduce of the flat gradient, 1/world folded into the Adam kernel -
runs on the real kernels.  (model.py:101-122 has no multi-GPU path; SURVEY 8(e) defines this one.)"""
import os
import socket
import warnings
from argparse import Namespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(seed, B=2, H=64, W=96, C=7):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, H, W, generator=g)
    y = torch.full((B, H, W), C, dtype=torch.int64)
    for b in range(B):
        idx = torch.randperm(H * W, generator=g)[:15]
        y[b].view(-1)[idx] = torch.randint(0, C, (15,), generator=g)
    return x.cuda(), y.cuda()


def _trainer(C=7):
    from pixelpick_amd.networks.layers import Dropout
    from pixelpick_amd.trainer import FlatTrainer
    from pixelpick_amd.utils.utils import get_model
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name="deeplab")).cuda().train()
    for mod in m.modules():
        if isinstance(mod, Dropout):
            mod.p = 0.0
    return FlatTrainer(m, ignore_index=C)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # (A) both ranks see the SAME batch: sum of two equal gradients x 1/2 is exact, so the parameters must equal a
        #     single-process run bit for bit
        tr = _trainer()
        assert tr.world == 2
        import pixelpick_amd.trainer as T
        assert T.OVERLAP_ALLREDUCE            # the decoder-side bucket is reduced under the encoder backward
        x, y = _batch(11)
        for _ in range(2):
            tr.train_step(x, y)
        pa = tr.flat_p.clone()
        ref = _trainer()
        ref.world, ref.collectives = 1, False           # no all-reduce, grad_scale 1
        for _ in range(2):
            ref.train_step(x, y)
        same_as_single = torch.equal(pa, ref.flat_p)
        # (B) disjoint shards: the replicas must stay identical to each other
        tr2 = _trainer()
        xs, ys = _batch(100 + rank)
        losses = [float(tr2.train_step(xs, ys)) for _ in range(2)]
        others = [torch.empty_like(tr2.flat_p) for _ in range(world)]
        dist.all_gather(others, tr2.flat_p)
        replicas_equal = all(torch.equal(others[0], o) for o in others[1:])
        differs_from_a = not torch.equal(tr2.flat_p, pa)
        torch.cuda.synchronize()
        # (C) the single all-reduce after backward (overlap off) gives the same parameters as the two-bucket overlap
        T.OVERLAP_ALLREDUCE = False
        tr3 = _trainer()
        for _ in range(2):
            tr3.train_step(xs, ys)
        T.OVERLAP_ALLREDUCE = True
        overlap_equals_plain = torch.equal(tr3.flat_p, tr2.flat_p)
        q.put((rank, same_as_single, replicas_equal, differs_from_a and overlap_equals_plain, losses))
    finally:
        dist.destroy_process_group()


def test_two_rank_train_step_on_one_gpu():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, same_as_single, replicas_equal, differs, losses in res:
        assert same_as_single, f"rank {rank}: identical shards must reproduce the single-process step"
        assert replicas_equal, f"rank {rank}: replicas diverged"
        assert differs and all(l == l and l < 50 for l in losses)


def _rccl_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        import pixelpick_amd.trainer as T
        x, y = _batch(11)
        ref = _trainer()
        assert not ref.collectives
        for _ in range(3):
            ref.train_step(x, y)
        out = {}
        for overlap in (True, False):
            T.FORCE_COLLECTIVES, T.OVERLAP_ALLREDUCE = True, overlap
            tr = _trainer()
            assert tr.collectives and tr.world == 1
            for _ in range(3):
                tr.train_step(x, y)
            torch.cuda.synchronize()
            out[overlap] = torch.equal(tr.flat_p, ref.flat_p)
            out[f"early{overlap}"] = hasattr(tr, "_comm_stream") == overlap
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_rccl_call_path_with_one_rank():
    """backend "nccl" (= RCCL) on the one GPU this box has: communicator setup, the overlapped decoder-side bucket issued
    from the helper stream and the encoder bucket after the join all run for real (PIXELPICK_FORCE_COLLECTIVES); a sum
    over one rank is the identity, so the parameters must equal the collective-free run bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert out[True] and out[False], out
    assert out["earlyTrue"] and out["earlyFalse"], out
