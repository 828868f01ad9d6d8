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


def _trainer(network="deeplab", C=7):
    from pixelpick_amd.networks.layers import Dropout
    from pixelpick_amd.trainer import FlatTrainer
    from pixelpick_amd.utils.utils import get_model
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name=network, weight_type="random", n_layers=50,
                                use_softmax=True, use_dilated_resnet=True, width_multiplier=1.0)).cuda().train()
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
        # three buckets for MobileNetV2: behind the encoder / its late blocks (both under the backward pass) / the first 0.9 MB
        assert 0 < tr2.n_mid < tr2.n_split < tr2.n and (tr2.n_split - tr2.n_mid) > 4 * tr2.n_mid
        tr2.time_collectives = True
        xs, ys = _batch(100 + rank)
        losses = [float(tr2.train_step(xs, ys)) for _ in range(2)]
        torch.cuda.synchronize()
        tags = [t for t, _, _ in tr2.comm_times]
        assert tags[:3] == ["behind_encoder", "encoder_late", "encoder_early"], tags
        tr2.time_collectives = False
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
        # (D) launch-plan replay under collectives: the recorded step carries both all-reduces (the early bucket on the helper
        #     stream included); three replayed steps equal three eager steps with device-side hyper-parameters
        tr4, tr5 = _trainer(), _trainer()
        tr4.enable_replay(xs, ys, warmup=0)
        for _ in range(2):
            tr4.train_step(xs, ys)
        for _ in range(3):
            tr5.step_count += 1
            tr5._stage_hyper()
            tr5._step_body(xs, ys, False, True)
        torch.cuda.synchronize()
        replay_equals_eager = torch.equal(tr4.flat_p, tr5.flat_p) and len(tr4._plan) > 100
        tr4.disable_replay()
        # (E) the ResNet50 encoder (FPNSeg): its late bucket is layer3 + layer4; three overlapped buckets == one all-reduce
        trf = _trainer("FPN")
        assert 0 < trf.n_mid < trf.n_split < trf.n and (trf.n_split - trf.n_mid) > 8 * trf.n_mid
        for _ in range(2):
            trf.train_step(xs, ys)
        T.OVERLAP_ALLREDUCE = False
        trg = _trainer("FPN")
        for _ in range(2):
            trg.train_step(xs, ys)
        T.OVERLAP_ALLREDUCE = True
        torch.cuda.synchronize()
        fpn_ok = torch.equal(trf.flat_p, trg.flat_p)
        q.put((rank, same_as_single, replicas_equal, differs_from_a and overlap_equals_plain and replay_equals_eager and fpn_ok, losses))
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
    os.environ["PIXELPICK_COMM_CU_RESERVE"] = "0"       # the reference run below has no collectives: same launch plans on both sides
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


# ----------------------------------------------------------------------------------------------------------------------
# Sharded acquisition round and the data-parallel driver: images i -> rank i mod W, one gather of the picks.
def _al_args(td, **kw):
    base = dict(dataset_name="cs", debug=False, dir_root=td, experim_name="dist", ignore_index=5, mc_n_steps=20,
                n_classes=5, n_pixels_by_us=10, network_name="deeplab", query_strategy="margin_sampling", reverse_order=False,
                stride_total=16, top_n_percent=0.0, use_mc_dropout=False, vote_type="hard", mc_dropout_p=0.2,
                n_init_pixels=10, max_budget=10, n_epochs=1, lr_scheduler_type="Poly", query_batch_size=2,
                optimizer_params={"lr": 5e-4, "betas": (0.9, 0.999), "weight_decay": 2e-4, "eps": 1e-7})
    base.update(kw)
    return Namespace(**base)


def _round(td, strategy, top, rev, model):
    import pickle
    import numpy as np
    from pixelpick_amd.query import QuerySelector
    from pixelpick_amd.synthetic import SyntheticDataset
    ds = SyntheticDataset(7, 64, 96, 5, 5, n_init_pixels=10, seed=3)          # 7 images: ragged over 2 ranks
    dl = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
    args = _al_args(td, query_strategy=strategy, top_n_percent=top, reverse_order=rev)
    qs = QuerySelector(args, dl, device=torch.device("cuda:0"))
    torch.manual_seed(77)
    np.random.seed(78)
    dq = qs(nth_query=1, model=model)
    stats = None
    p = f"{td}/checkpoints/dist/1_query/query_stats.pkl"
    if os.path.exists(p):
        stats = pickle.load(open(p, "rb"))
    return dq, stats, [q.copy() for q in ds.queries], ds.n_getitem


def _acq_worker(rank, world, port, q, td):
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    warnings.simplefilter("ignore")
    from pixelpick_amd.utils.utils import get_model
    torch.manual_seed(0)
    model = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=5, network_name="deeplab")).cuda()
    cases = [("margin_sampling", 0.0, False), ("entropy", 0.05, False), ("least_confidence", 0.05, True), ("random", 0.0, False)]
    single = [_round(f"{td}/single{rank}_{i}", *c, model) for i, c in enumerate(cases)]           # torch.distributed not initialised
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = []
        for i, c in enumerate(cases):
            dq, stats, queries, n_loaded = _round(f"{td}/sharded_{i}", *c, model)          # rank 0 writes the statistics file
            dist.barrier()
            dq1, stats1, queries1, n_loaded1 = single[i]
            same = list(dq.keys()) == list(dq1.keys())
            # host side of the shard: this rank's loader collated ceil((7 - rank) / 2) images, the single-rank round all 7
            same = same and n_loaded1 == 7 and n_loaded == len(range(rank, 7, world))
            for k in dq1:
                same = same and np.array_equal(dq[k]["x_coords"], dq1[k]["x_coords"]) and np.array_equal(dq[k]["y_coords"], dq1[k]["y_coords"])
            same = same and all(np.array_equal(a, b) for a, b in zip(queries, queries1))    # label_queries side effect on every rank
            if rank == 0:
                same = same and stats is not None and stats["label_distribution"] == stats1["label_distribution"]
                for key in ("avg_n_unique_labels", "avg_spatial_coverage"):
                    same = same and stats[key] == stats1[key]                      # bit-identical: merged in loader order
                # entropies: an image forwarded in a batch of another size goes through other conv tilings (1e-7 on the logits)
                same = same and abs(stats["avg_entropy"] - stats1["avg_entropy"]) <= 1e-5 * abs(stats1["avg_entropy"])
            ok.append(bool(same))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_sharded_acquisition_round_equals_single_rank(tmp_path):
    """Two ranks (gloo, sharing cuda:0) run QuerySelector.__call__ on the same 7-image dataset: dict_queries, the dataset's
    merged masks on EVERY rank and rank 0's query_stats.pkl must equal the single-rank round (coordinates, counts and
    coverage bit for bit; the mean entropy to 1e-5: batch composition changes the conv tiling) - for a plain
    top-k strategy, the top-5 % + numpy sub-sample mode, reverse-order sampling and the host-RNG `random` strategy."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_acq_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok in res:
        assert all(ok), f"rank {rank}: {ok}"


def _driver_worker(rank, world, port, q, td):
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    warnings.simplefilter("ignore")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pixelpick_amd.model import Model
        from pixelpick_amd.synthetic import SyntheticDataset
        torch.manual_seed(5)                                   # same seed on every rank: same shuffle order, disjoint shards
        np.random.seed(5)
        class FileWritingDataset(SyntheticDataset):
            """label_queries dumps queries.pkl whenever nth_query is an int, as the reference's datasets do (base_dataset.py:43-45)."""
            writes = 0

            def label_queries(self, queries, nth_query=None):
                super().label_queries(queries, nth_query)
                if isinstance(nth_query, int):
                    import pickle
                    os.makedirs(f"{td}/checkpoints/dist/{nth_query}_query", exist_ok=True)
                    with open(f"{td}/checkpoints/dist/{nth_query}_query/queries.pkl", "wb") as f:
                        pickle.dump(queries, f)
                    FileWritingDataset.writes += 1

        ds = FileWritingDataset(8, 64, 96, 5, 5, n_init_pixels=10, seed=1)
        ds_q = FileWritingDataset(8, 64, 96, 5, 5, n_init_pixels=10, seed=1)
        ds_val = SyntheticDataset(3, 64, 96, 5, 5, seed=2)
        mk = lambda d, b, sh: torch.utils.data.DataLoader(d, batch_size=b, shuffle=sh)
        m = Model(_al_args(td), mk(ds, 2, True), mk(ds_q, 1, False), mk(ds_val, 1, False), device=torch.device("cuda:0"))
        assert m.world == 2
        m()
        # only rank 0 wrote queries.pkl (2 stages x (query dataset + train dataset)); the other rank merged in memory
        assert FileWritingDataset.writes == (4 if rank == 0 else 0), FileWritingDataset.writes
        import pickle
        assert len(pickle.load(open(f"{td}/checkpoints/dist/1_query/queries.pkl", "rb"))) == 8      # complete file on every rank's view
        # host side of the shards: per stage this rank collated 8/2 train images per epoch, 4 query images, ceil((3-rank)/2) val images
        assert ds.n_getitem == 2 * 4 and ds_q.n_getitem == 2 * 4, (ds.n_getitem, ds_q.n_getitem)
        assert ds_val.n_getitem == 2 * len(range(rank, 3, world)), ds_val.n_getitem
        picks = np.stack([qq for qq in ds.queries]).astype(np.uint8)
        t = torch.from_numpy(picks)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        same_masks = all(torch.equal(parts[0], p) for p in parts[1:])
        n_per_image = [int(qq.sum()) for qq in ds.queries]
        files = sorted(os.listdir(f"{td}/checkpoints/dist/0_query")) if rank == 0 else None
        q.put((rank, same_masks, n_per_image, [h for h in m.history if h[0] == "train"], files))
    finally:
        dist.destroy_process_group()


def test_two_rank_driver_keeps_replicas_and_datasets_identical(tmp_path):
    """Model(args)() on two ranks: sharded train batches with the gradient all-reduce, sharded validation with summed
    confusion matrices, sharded acquisition; both ranks must end every stage with the same labelled masks and the same
    logged scores, and only rank 0 writes the checkpoint / log / statistics files."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_driver_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert res[0][2] == res[1][2] == [10 + 2 * 10] * 8            # initial 10 + two acquisition rounds of 10
    assert res[0][3] == res[1][3] and len(res[0][3]) == 2           # identical train history (summed confusion matrices)
    assert {"best_miou_model.pt", "log_train.txt", "log_val.txt", "query_stats.pkl"} <= set(res[0][4])


def _long_worker(rank, world, port, q, steps):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pixelpick_amd.trainer as T
        from pixelpick_amd import engine as E
        assert T.OVERLAP_ALLREDUCE and E.Tape.overlap_wgrad and E._BN_FUSED
        from pixelpick_amd.utils.utils import get_model
        from pixelpick_amd.trainer import FlatTrainer
        torch.manual_seed(rank)                                  # DIFFERENT initial weights per rank: the constructor must
        with warnings.catch_warnings():                          # broadcast rank 0's
            warnings.simplefilter("ignore")
            m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=7, network_name="deeplab")).cuda().train()
        tr = FlatTrainer(m, ignore_index=7)                      # dropout ON (p = 0.5 / 0.2): per-rank masks, same parameters
        E.set_dropout_seed(1000 + rank)
        batches = [_batch(300 + 10 * rank + i) for i in range(4)]
        checks = []
        for s in range(steps):
            tr.train_step(*batches[s % 4])
            if (s + 1) % 50 == 0:
                others = [torch.empty_like(tr.flat_p) for _ in range(world)]
                dist.all_gather(others, tr.flat_p)
                checks.append(all(torch.equal(others[0], o) for o in others[1:]))
        torch.cuda.synchronize()
        q.put((rank, checks, bool(torch.isfinite(tr.flat_p).all().item()), float(tr.last_loss.item())))
    finally:
        dist.destroy_process_group()


def test_two_ranks_200_steps_with_spin_barrier_batchnorm_side_stream_and_overlapped_buckets():
    """200 optimisation steps on two ranks sharing one GPU (gloo): the single-launch (spin-waiting) BatchNorm of BOTH
    processes, the weight-gradient side streams and the overlapped two-bucket all-reduce all run concurrently on the same
    device - the co-residency cap (pp_bn_fused_capacity / 2 per launch) must keep that from hanging - and the replicas,
    started from different random initialisations, stay bit-identical (checked every 50 steps)."""
    world, steps = 2, 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_long_worker, args=(r, world, port, q, steps)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, checks, finite, loss in res:
        assert len(checks) == 4 and all(checks), f"rank {rank}: replicas diverged {checks}"
        assert finite and loss == loss


def _occupied_worker(rank, world, port, q, cus_list, steps):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes
        from pixelpick_amd import _lib
        from pixelpick_amd import engine as E
        from pixelpick_amd.trainer import FlatTrainer
        from pixelpick_amd.utils.utils import get_model
        L = _lib.lib()
        out = []
        occ_stream = torch.cuda.Stream()
        stop = torch.zeros(1, dtype=torch.int32).pin_memory()
        started = torch.zeros(1, dtype=torch.int64, device="cuda")
        g = torch.Generator().manual_seed(500 + rank)
        x = torch.randn(4, 3, 256, 512, generator=g).cuda()
        y = torch.full((4, 256, 512), 19, dtype=torch.int64)
        for b in range(4):
            idx = torch.randperm(256 * 512, generator=g)[:20]
            y[b].view(-1)[idx] = torch.randint(0, 19, (20,), generator=g)
        y = y.cuda()
        for cus in cus_list:
            os.environ["PIXELPICK_COMM_CU_RESERVE"] = str(cus)          # what a data-parallel run under RCCL sets (default 32 there)
            torch.manual_seed(0)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=19, network_name="deeplab")).cuda().train()
            tr = FlatTrainer(m, ignore_index=19)
            assert L.pp_get_comm_cu_reserve() == cus
            E.set_dropout_seed(77 + rank)
            tr.train_step(x, y)                                          # scratch buffers grow before the occupier sits down
            torch.cuda.synchronize()
            dist.barrier()
            n_started = 0
            if rank == 0:
                # the stand-in for RCCL's channel blocks: `cus` blocks that each own a whole CU's LDS for the whole run (at most 40 s)
                stop[0] = 0
                started.zero_()
                torch.cuda.synchronize()
                _lib.check(L.pp_debug_occupy_cus(cus, stop.data_ptr(), 4_000_000_000, started.data_ptr(), occ_stream.cuda_stream), "occupy")
                for _ in range(2000):
                    n_started = int(started.cpu().item())
                    if n_started == cus:
                        break
            dist.barrier()
            for _ in range(steps):
                tr.train_step(x, y)
            torch.cuda.current_stream().synchronize()                    # (not the device: the occupier is still spinning)
            others = [torch.empty_like(tr.flat_p) for _ in range(world)]
            dist.all_gather(others, tr.flat_p)
            same = all(torch.equal(others[0], o) for o in others[1:])
            dist.barrier()
            if rank == 0:
                stop[0] = 1
                occ_stream.synchronize()
            dist.barrier()
            out.append((cus, n_started, same, bool(torch.isfinite(tr.flat_p).all().item())))
            del tr, m
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_spin_wait_launches_survive_a_resident_occupier_kernel():
    """with RCCL, channel blocks sit on CUs for the whole of an overlapped all-reduce while the single-launch
    BatchNorm / convolution + BatchNorm grids wait for co-resident siblings.  Stand-in: an occupier kernel parked on 16 / 32 / 64
    CUs (every block owns a CU's whole LDS) on a third stream for the whole of a run of two-rank steps at the BASELINE shape
    (B = 4, 256 x 512; both processes' spin-waiting launches, side streams and the overlapped buckets concurrent on ONE device),
    with PIXELPICK_COMM_CU_RESERVE = the occupied CUs.  Nothing may hang, the occupier must really have been resident, and the
    replicas stay bit-identical."""
    world, steps, cus_list = 2, 60, (16, 32, 64)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_occupied_worker, args=(r, world, port, q, cus_list, steps)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, out in res:
        assert [o[0] for o in out] == list(cus_list)
        for cus, n_started, same, finite in out:
            assert same and finite, (rank, cus)
            if rank == 0:
                assert n_started == cus, f"only {n_started} of {cus} occupier blocks were resident"


def test_bench_line_from_two_ranks_through_torch_distributed_run():
    """The driver's own launch line for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py
    --gpus N`), with two ranks sharing this box's one GPU over gloo (PIXELPICK_DIST_BACKEND: RCCL needs a device per rank):
    rank 0 prints ONE JSON line whose value is the whole-job aggregate, with the distributed block filled in."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PIXELPICK_DIST_BACKEND="gloo", OMP_NUM_THREADS="4")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "8", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_batch"] == 8                       # per-GPU batch 4 x 2 ranks
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    dd = d["distributed"]
    assert dd["nranks"] == 2 and dd["backend"] == "gloo" and dd["allreduce_bytes_per_step"] == 4 * 5815539 and sum(dd["buckets"]) == 4 * 5815539
    assert "roofline" in d and d["acquisition"]["value"] > 0


@pytest.mark.parametrize("replay,ranks", [("off", 2), ("on", 2), ("on", 8)])
def test_bench_gpus_n_without_a_launcher_spawns_its_own_ranks(replay, ranks):
    """`python bench.py --gpus 2` typed WITHOUT torch.distributed.run re-launches itself through the driver's line (one wrong
    launch line used to end in an assertion): same single JSON line, plus the host enqueue time per step, whether the
    launch-plan replay was on, and what each gradient bucket cost INSIDE the step."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PIXELPICK_DIST_BACKEND="gloo", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    # ranks = 8: the driver's `--gpus 8` line with eight ranks sharing this box's one GPU over gloo - plumbing only (nranks, buckets,
    # the global batch; the replayed step returns to Python for each of the three all-reduces)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1", "--batch", "8",
           "--no-cpu-baseline", "--mode", "train", "--replay", replay]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["value"] > 0 and d["config"]["global_batch"] == 4 * ranks
    # `value` is the WHOLE-JOB aggregate: the images of all ranks (per-GPU batch 4 each) over the max-over-ranks step time
    assert abs(d["value"] - 4 * ranks / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    t = d["train"]
    assert t["replay"] == (replay == "on") and 0 < t["host_enqueue_ms_per_step"] < 1e3 and t["host_cores_per_rank"] > 0
    dd = d["distributed"]
    assert dd["nranks"] == ranks and set(dd["allreduce_in_step_us"]) == {"behind_encoder", "encoder_late", "encoder_early"}
    assert len(dd["buckets"]) == 3 and sum(dd["buckets"]) == dd["allreduce_bytes_per_step"] and dd["buckets"][2] < dd["buckets"][1]
    assert all(v > 0 for v in dd["allreduce_in_step_us"].values())
    # the first-real-node diagnostics: every rank's device, the shared-device flag (all ranks sit on this box's one GPU), the comm-CU
    # reserve beside the communicator's channels (gloo: none), and the N = 1 figure of the same invocation
    assert [r["rank"] for r in dd["rank_devices"]] == list(range(ranks)) and all(r["device"] == 0 for r in dd["rank_devices"])
    assert dd["shared_device"] is True and dd["devices_visible"] == 1
    assert dd["comm_cu_reserve"] == 0 and dd["rccl_channels"] is None and dd["rccl_channels_source"] == "not an RCCL communicator"
    n1 = dd["n1_same_invocation"]
    assert n1["img_per_s"] > 0 and n1["replay"] == (replay == "on") and abs(n1["img_per_s"] - 4 / (n1["ms_per_step"] * 1e-3)) < 0.02 * n1["img_per_s"]
    assert 0 < dd["scaling_vs_n1_same_invocation"] <= 1.05          # ranks sharing one device cannot beat one rank alone
