#!/usr/bin/env python3
"""bench.py — PixelPick hot path on MI355X: ONE JSON line per run (driver contract).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

BASELINE.json metric: images/s of a full DeepLabv3+-MobileNetV2 train step + Mpixels/s of acquisition, at
Cityscapes-quarter 256x512 (configs[1]).  One run measures both on synthetic data resident in HBM:

  value / ms_per_step   K train steps (model.py:101-122: forward, sparse cross-entropy over 20 labelled
                        pixels/image, backward, [RCCL all-reduce of the flat gradient], Adam with the backbone
                        at lr/10), per-GPU batch 4 (args.py:89), BN in train mode, dropout active, fp32.
                        One process per GPU; value = images of ALL ranks / max-over-ranks time ("weak").
  acquisition{}         K passes of softmax -> entropy -> exclusion -> per-image top-20 (query.py:190-204,
                        57-61) over B=256 images of 256x512x19 logits per GPU (2.58 GB >> 256 MiB MALL);
                        images shard over ranks, no collective.
  roofline{}            the roofline-graded kernel (north_star): acq_kernel vs HBM.  achieved = algorithmic
                        bytes per launch pixels*(4C+1) / mean launch duration from HIP events recorded around
                        that kernel on its stream, inside the timed region.
  roofline_mfma{}       the dominant train-step kernel (conv_igemm_kernel, SegmentHead 3x3 304->256 at
                        B*64*128 rows, decoders.py:107) vs the fp32 MFMA peak, same event method.
  roofline_mfma_1x1{}   the SURVEY 8(d) graded 1x1 (pointwise) shapes at the BASELINE batch, each vs the fp32 MFMA peak
                        and vs its own HBM-limited ceiling.
  cpu_baseline{}        the plain-PyTorch port of the same train step / acquisition loop (oracle/) timed on
                        this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time
import warnings
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TF = 2516.6 # MI355X_MICROARCH.md: dense bf16 MFMA peak (16x the fp32 MFMA rate)


class HipEvents:
    """n start/stop hipEvent pairs created through the HIP runtime PyTorch already loaded."""

    def __init__(self, n):
        self.hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.n = n
        self.starts = (ctypes.c_void_p * n)()
        self.stops = (ctypes.c_void_p * n)()
        for arr in (self.starts, self.stops):
            for i in range(n):
                ev = ctypes.c_void_p()
                rc = self.hip.hipEventCreate(ctypes.byref(ev))
                assert rc == 0, f"hipEventCreate -> {rc}"
                arr[i] = ev

    def elapsed_ms(self):
        out = []
        for i in range(self.n):
            ms = ctypes.c_float()
            rc = self.hip.hipEventElapsedTime(ctypes.byref(ms), self.starts[i], self.stops[i])
            assert rc == 0, f"hipEventElapsedTime -> {rc}"
            out.append(ms.value)
        return out

    def destroy(self):
        for arr in (self.starts, self.stops):
            for i in range(self.n):
                self.hip.hipEventDestroy(arr[i])


# ------------------------------------------------------------------------------------------------ synthetic inputs
def synth_train_batch(B, C, H, W, n_lab, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn((B, 3, H, W), device=device, generator=g)
    y = torch.full((B, H, W), C, dtype=torch.int64, device=device)          # all ignore_index (= C) ...
    for b in range(B):                                                      # ... except n_lab labelled pixels/img
        idx = torch.randperm(H * W, device=device, generator=g)[:n_lab]
        y[b].view(-1)[idx] = torch.randint(0, C, (n_lab,), device=device, generator=g)
    return x, y


# ------------------------------------------------------------------------------------------------ CPU baselines (oracle/)
def cpu_baseline_train(B, C, H, W, n_lab, budget_s=15.0):
    from oracle import net as onet
    threads = torch.get_num_threads()
    torch.manual_seed(0)
    model = onet.OracleDeepLab(C).train()
    opt = onet.make_optimizer(model)
    x, y = synth_train_batch(B, C, H, W, n_lab, torch.device("cpu"), 1)
    onet.train_step(model, opt, x, y, C)                                     # warm-up
    t0 = time.perf_counter()
    done = 0
    while True:
        onet.train_step(model, opt, x, y, C)
        done += 1
        el = time.perf_counter() - t0
        if el > budget_s or done >= 50:
            break
    return round(done * B / el, 3), f"{done} train steps of B={B} {H}x{W} (oracle/net.py, torch CPU ops, {threads} threads, {el:.1f} s)", threads


def cpu_baseline_acq(C, H, W, k, strategy, budget_s=8.0):
    from oracle import acq as orc
    gen = torch.Generator().manual_seed(0)
    n_img = 4
    logits = [torch.randn(1, C, H, W, generator=gen) * 3 for _ in range(n_img)]
    excl = [torch.rand(H, W, generator=gen) < 0.05 for _ in range(n_img)]
    orc.torch_port_acquire(logits[0], excl[0], strategy, k)
    t0 = time.perf_counter()
    done = 0
    while True:
        orc.torch_port_acquire(logits[done % n_img], excl[done % n_img], strategy, k)
        done += 1
        el = time.perf_counter() - t0
        if el > budget_s or done >= 2000:
            break
    return round(done * H * W / el / 1e6, 3), f"{done} images {H}x{W}x{C} one at a time (query.py:159 loop), {el:.1f} s"


def train_step_cost_model(tr, x, y):
    """The floors one train step of THIS network cannot beat, from one instrumented eager forward (nothing is timed here):
      flops   dense convolutions only: 2*M*Cin*Cout*kh*kw for the forward, the same again for backward-data (where the input needs
              a gradient) and for the weight gradient; layers the planner puts on the bf16x3 path (engine._x3_planes: six bf16 MFMAs per
              fp32 product, ceiling 2516.6 / 6 = 419.4 TF) are priced at that ceiling, all others at the fp32 MFMA peak (157.3 TF);
      bytes   every tape node (convolution, depthwise, BatchNorm, GroupNorm, pooling, interpolation, add, dropout ...) moves at least
              its inputs + its output in the forward and d(output) + saved inputs + d(inputs) in the backward: 3*in + 2*out fp32
              words; zero-copy nodes (concat views) move nothing; + 2 reads of every weight, one write of its gradient and the
              optimiser's 4 reads + 3 writes per parameter.  No reuse through L2 / MALL is assumed away: it is a floor on HBM bytes
              only if nothing stays on chip between producer and consumer, i.e. an UPPER estimate of the unavoidable traffic."""
    from pixelpick_amd import engine as E
    convs, nodes = [], []
    orig_conv, orig_record = E.conv2d, E.Tape.record

    def numel(v):
        try:
            shp = E.shape_of(v)
        except Exception:
            shp = tuple(v.t.shape)
        n = 1
        for d in shp:
            n *= int(d)
        return n

    def rec_conv(tape, xv, w, bias, stride=1, pad=0, dil=1, dst=None, **kw):
        B, H, W, Cin = E.shape_of(xv)
        convs.append((B, H, W, Cin, int(w.shape[3]), int(w.shape[0]), int(w.shape[1]), stride, pad, dil, bool(xv.needs_grad)))
        return orig_conv(tape, xv, w, bias, stride, pad, dil, dst, **kw)

    def rec_record(self, fn, ctx, out):
        if self.enabled and out is not None and getattr(fn, "__name__", "") != "_concat_bwd":
            n_in = 0
            for c in ctx:
                for v in (c if isinstance(c, (list, tuple)) else (c,)):
                    if isinstance(v, E.Var):
                        n_in += numel(v)
            nodes.append((getattr(fn, "__name__", "?"), n_in, numel(out)))
        return orig_record(self, fn, ctx, out)

    patched = []
    E.conv2d = rec_conv
    for mod in list(sys.modules.values()):
        if mod and getattr(mod, "__name__", "").startswith("pixelpick_amd") and getattr(mod, "conv2d", None) is orig_conv:
            mod.conv2d = rec_conv
            patched.append(mod)
    E.Tape.record = rec_record
    try:
        tr.forward_backward(x, y)
    finally:
        E.conv2d = orig_conv
        for mod in patched:
            mod.conv2d = orig_conv
        E.Tape.record = orig_record
    fl_x3 = fl_f32 = 0.0
    for (B, H, W, Cin, Cout, kh, kw, s_, p_, d_, ng) in convs:
        Ho, Wo = E.out_size(H, kh, s_, p_, d_), E.out_size(W, kw, s_, p_, d_)
        f = 2.0 * B * Ho * Wo * Cin * Cout * kh * kw
        for which, on in ((0, True), (1, ng), (2, True)):
            if not on:
                continue
            if E._x3_planes(which, B, H, W, Cin, Cout, kh, kw, s_, p_, d_):
                fl_x3 += f
            else:
                fl_f32 += f
    act_bytes = 4.0 * sum(3 * n_in + 2 * n_out for _, n_in, n_out in nodes)
    par_bytes = 4.0 * tr.n * (2 + 1 + 7)
    mfma_floor = fl_x3 / (MFMA_BF16_PEAK_TF / 6 * 1e12) + fl_f32 / (MFMA_F32_PEAK_TF * 1e12)
    hbm_floor = (act_bytes + par_bytes) / (HBM_PEAK_GBS * 1e9)
    return {"flops_per_step": fl_x3 + fl_f32, "flops_on_bf16x3_pipe": fl_x3, "flops_on_fp32_pipe": fl_f32, "dense_conv_layers": len(convs),
            "mfma_floor_ms": round(mfma_floor * 1e3, 4), "hbm_bytes_per_step": act_bytes + par_bytes, "tape_nodes": len(nodes),
            "hbm_floor_ms": round(hbm_floor * 1e3, 4),
            "model": "flops: dense convolutions fwd + bwd-data + bwd-weight, bf16x3-planned layers at 419.4 TF, the rest at the 157.3 TF fp32 MFMA "
                     "peak; bytes: 3*in + 2*out fp32 words per tape node, 10 words per parameter, at 8 TB/s (bench.py:train_step_cost_model)"}


def cpu_baseline_grad_deviation(C, H, W, B, n_lab, dev):
    """Checker leg (cpu_baseline block, untimed): ONE free-running forward / backward of the HIP DeepLabv3+-MNv2 and of oracle/net.py
    from the same weights (dropout off in both), same batch: relative L2 deviation of every parameter gradient.  tests/
    test_layerwise_parity_gpu.py holds the per-layer (forced) and branch-aligned bars; this is the free-running number the
    review asked to see every round - it is dominated by ReLU units whose pre-activation sits within rounding of 0 and takes the
    other branch, not by kernel error."""
    from oracle import net as onet
    from pixelpick_amd.networks.layers import Dropout
    from pixelpick_amd.trainer import FlatTrainer
    from pixelpick_amd.utils.utils import get_model
    torch.manual_seed(7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name="deeplab", weight_type="random"))
    for mod in m.modules():
        if isinstance(mod, Dropout):
            mod.p = 0.0
    o = onet.OracleDeepLab(C, 0.0, 0.0, 0.0)
    o.load_state_dict(m.state_dict())
    x, y = synth_train_batch(B, C, H, W, n_lab, torch.device("cpu"), 11)
    lo = torch.nn.functional.cross_entropy(o.train()(x), y, ignore_index=C)
    lo.backward()
    m = m.to(dev).train()
    tr = FlatTrainer(m, ignore_index=C)
    lh = tr.forward_backward(x.to(dev), y.to(dev))
    torch.cuda.synchronize(dev)
    og = dict(o.named_parameters())
    rel, norms = [], []
    for k, p_ in m.named_parameters():
        g = tr._grad_view[id(p_)].detach().cpu()
        if g.dim() == 4:
            g = g.permute(3, 2, 0, 1)
        elif g.dim() == 3:
            g = g.permute(2, 0, 1).unsqueeze(1)
        r = og[k].grad
        nr = float(r.double().norm())
        rel.append((float((g.double() - r.double()).norm() / max(nr, 1e-30)), nr, k))
        norms.append(nr)
    # gradients that are analytically ~0 (the beta of a BatchNorm whose consumers are all conv -> BatchNorm: a cancellation residue
    # 1e-6 of the typical gradient norm) say nothing as a ratio; they are counted apart, as tests/layerwise.py does
    typical = sorted(norms)[len(norms) // 2]
    resid = [t for t in rel if t[1] < 1e-4 * typical]
    rel = sorted(t for t in rel if t[1] >= 1e-4 * typical)
    del tr, m
    torch.cuda.empty_cache()
    return {"tensors": len(rel), "rel_l2_max": float(f"{rel[-1][0]:.3e}"), "rel_l2_max_tensor": rel[-1][2],
            "rel_l2_median": float(f"{rel[len(rel) // 2][0]:.3e}"),
            "analytically_zero_gradients_excluded": len(resid),
            "loss_hip": round(float(lh.item()), 6), "loss_oracle": round(float(lo.item()), 6),
            "what": f"free-running fwd + bwd of B={B} {H}x{W}, {n_lab} labelled px/img, HIP vs oracle/net.py from the same weights; "
                    "bars: tests/test_layerwise_parity_gpu.py (forced 2e-4, branch-aligned 5e-4)"}


def cpu_baseline_pick_flips(dev):
    """Checker leg (outside every timed region, rank 0 at N = 1 only, part of the cpu_baseline block): the picks of the HIP path
    with the default scorer (algebraic entropy form, v_exp_f32) and with the reference's operation order
    (strategy | PP_ACQ_REFERENCE_ORDER: p = exp(x - m) / S, sum(-p log p), query.py:190,229-239) against oracle/acq_oracle.c on UNGUARDED
    random logits - how often a device pick differs from the oracle's (a k-th / (k+1)-th score pair closer than the rounding
    difference of the two evaluations).  8 images of 256x512x19, 8 of 320x320x21, one 1024x2048x19; k = 20."""
    from oracle import acq as orc
    from pixelpick_amd import _lib
    from pixelpick_amd import acquisition as acq
    L = _lib.lib()
    gen = torch.Generator(device=dev).manual_seed(4242)
    out = {"k": 20, "what": "device picks vs oracle/acq_oracle.c (reference operation order, libm) on unguarded randn*3 logits; "
                          "set_flips = picks not in the oracle's set, order_flips = positions of the value-sorted list that differ"}
    for label, B, C, H, W, strategies in (("256x512x19", 8, 19, 256, 512, ("entropy", "least_confidence", "margin_sampling")),
                                          ("320x320x21", 8, 21, 320, 320, ("margin_sampling", "entropy")),
                                          ("1024x2048x19", 1, 19, 1024, 2048, ("least_confidence", "entropy"))):
        logits = torch.randn((B, C, H, W), device=dev, generator=gen) * 3
        excl = (torch.rand((B, H, W), device=dev, generator=gen) < 0.05).to(torch.uint8)
        lg_np, ex_np = logits.cpu().numpy(), excl.cpu().numpy()
        for st in strategies:
            o_idx, _ = orc.score_topk(lg_np, ex_np, st, 20)
            rec = {"images": B, "picks": B * 20}
            for name, exact in (("default", False), ("exact_formula", True)):
                idx, _, _ = acq.score_topk(logits, excl, st, 20, reference_order=exact)
                d = idx.cpu().numpy()
                rec[name] = {"set_flips": int(sum(len(set(d[b].tolist()) - set(o_idx[b].tolist())) for b in range(B))),
                             "order_flips": int((d != o_idx).sum())}
            out[f"{label} {st}"] = rec
        del logits, excl
    torch.cuda.empty_cache()
    return out


def _acq_source_hash():
    import hashlib
    h = hashlib.sha256()
    here = os.path.dirname(os.path.abspath(__file__))
    for f in ("acq.hip", "pp_common.h"):
        h.update(open(os.path.join(here, "pixelpick_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--train-batch", type=int, default=4, help="images per GPU per train step (args.py:89)")
    ap.add_argument("--batch", type=int, default=256, help="acquisition: images per launch per GPU")
    ap.add_argument("--classes", type=int, default=19)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--n-labelled", type=int, default=20)
    ap.add_argument("--strategy", default="entropy")
    ap.add_argument("--layout", default="nchw", choices=["nchw", "nhwc"])
    ap.add_argument("--mode", default="both", choices=["both", "train", "acq"])
    ap.add_argument("--network", default="deeplab", choices=["deeplab", "FPN", "deeplab_r50"],
                    help="train leg: deeplab = DeepLabv3+-MobileNetV2 (BASELINE configs[1], the default line); FPN = the "
                         "reference's ResNet50 model (networks/model.py FPNSeg, configs[2]); deeplab_r50 = the DeepLabv3+-ResNet50 "
                         "assembled from the reference's parts (SURVEY.md 0.1, an extra: the reference never builds it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the sub-records of the other BASELINE configurations")
    ap.add_argument("--tune-occ", type=int, default=0)
    ap.add_argument("--tune-ppt", type=int, default=0)
    ap.add_argument("--reduce-mode", type=int, default=0)
    ap.add_argument("--exact-formula", type=int, default=0)
    ap.add_argument("--conv-variant", type=int, default=0)
    ap.add_argument("--knobs-build", action="store_true",
                    help="measure libpixelpick_hip_knobs.so (the test build: same sources + the pp_debug_* planner switches) instead of the "
                         "product library; needed by --tune-occ / --tune-ppt / --reduce-mode / --conv-variant and by the fp32-MFMA A/B sub-record")
    ap.add_argument("--replay", default="auto", choices=["auto", "on", "off"],
                    help="re-issue the train step's recorded launch list (FlatTrainer.enable_replay: same GPU schedule, half the host "
                         "time per step).  auto: on when this process has fewer than 8 host cores per rank to enqueue from")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the driver's launch line (one rank per GPU over RCCL), same output
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    rccl_log = None
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 or world > 1:
        import torch.distributed as dist
        assert world == a.gpus, f"WORLD_SIZE={world} but --gpus {a.gpus}: launch with torch.distributed.run"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # PIXELPICK_DIST_BACKEND=gloo lets the N>1 code path run on a box with fewer GPUs than ranks (ranks then share
        # devices; RCCL needs one device per rank) - a plumbing check, not a measurement
        backend = os.environ.get("PIXELPICK_DIST_BACKEND", "nccl")
        dev_index = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        if backend == "nccl" and "NCCL_DEBUG" not in os.environ:
            # the communicator's INIT lines (channel count) go to a scratch file per rank: bench.py reports `rccl_channels` from rank 0's
            import tempfile
            rccl_log = os.path.join(tempfile.gettempdir(), f"pixelpick_rccl_{os.getpid()}.log")
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT", NCCL_DEBUG_FILE=rccl_log)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)

    from pixelpick_amd import _lib
    from pixelpick_amd import acquisition as acq
    from pixelpick_amd import dist_utils as du
    if a.knobs_build:
        _lib.use_knobs_build()
    L = _lib.lib()
    KNOBS = _lib.knobs_build()
    if KNOBS:
        L.pp_debug_set_acq_tuning(a.tune_occ, a.tune_ppt)
        L.pp_debug_set_reduce_mode(a.reduce_mode)
        L.pp_debug_set_conv_variant(a.conv_variant)
    else:
        assert not (a.tune_occ or a.tune_ppt or a.reduce_mode or a.conv_variant), "planner switches exist in the test build only: add --knobs-build"
    # the reference's operation order is a per-call flag of the product library (PP_ACQ_REFERENCE_ORDER), not a process switch
    exact_flag = [0x100 if a.exact_formula else 0]
    stream = torch.cuda.current_stream(dev).cuda_stream
    C, H, W, k = a.classes, a.height, a.width, a.k

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(el):
        if dist is None:
            return el
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(step_fn, steps, warmup, events=None):
        for _ in range(warmup):
            step_fn()
        barrier()
        if events is not None:
            L.pp_set_kernel_events(events.starts, events.stops, events.n)
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        barrier()
        el = time.perf_counter() - t0
        L.pp_set_kernel_events(None, None, 0)
        return max_over_ranks(el)

    line = {}
    library_build = "test build (libpixelpick_hip_knobs.so)" if KNOBS else "product (libpixelpick_hip.so: no pp_debug_* switches)"

    # ------------------------------------------------------------------------------------ train step
    def train_leg(network, steps, warmup, Ht, Wt, Ct, headline=False):
        """K train steps of `network` at per-GPU batch --train-batch on [Ht, Wt] crops -> (record, trainer, model)."""
        from pixelpick_amd.trainer import FlatTrainer
        from pixelpick_amd.utils.utils import get_model
        from pixelpick_amd import engine as E
        TB = a.train_batch
        torch.manual_seed(0)                         # identical replicas on every rank
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=Ct, network_name=network,
                                        weight_type="random", n_layers=50, use_softmax=True, use_dilated_resnet=True,
                                        width_multiplier=1.0)).to(dev).train()
        tr = FlatTrainer(model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=2e-4, ignore_index=Ct)
        E.set_dropout_seed(1234 + rank)
        x, y = synth_train_batch(TB, Ct, Ht, Wt, a.n_labelled, dev, 1 + rank)       # disjoint shards per rank
        cores_per_rank = (os.cpu_count() or 1) / max(world, 1)
        replay = a.replay == "on" or (a.replay == "auto" and cores_per_rank < 8)
        replay_why = "forced" if a.replay == "on" else ("fewer than 8 host cores per rank" if replay else None)
        if a.replay == "auto" and not replay:
            # probe (untimed, before the warmup): what a step costs the HOST when issued into empty queues against what it costs the
            # GPU.  A host that needs more than 0.9 of the GPU step to enqueue it would bound the run on a busier or slower box than
            # this one: take the recorded launch list then (about half the host cost, same GPU schedule, bit-identical steps).
            for _ in range(3):
                tr.train_step(x, y)
            probe = []
            for _ in range(3):
                torch.cuda.synchronize(dev)
                t = time.perf_counter()
                tr.train_step(x, y)
                probe.append(time.perf_counter() - t)
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            for _ in range(5):
                tr.train_step(x, y)
            torch.cuda.synchronize(dev)
            gpu_step = max_over_ranks((time.perf_counter() - t) / 5)
            host_step = max_over_ranks(sorted(probe)[1])
            if host_step > 0.9 * gpu_step:       # (the replayed step is ~1 % slower on the GPU: only when the host really is the limit)
                replay, replay_why = True, f"host enqueue {host_step * 1e3:.2f} ms > 0.9 x step {gpu_step * 1e3:.2f} ms in the probe"
        if replay:
            tr.enable_replay(x, y, warmup=1)         # recorded launch list, eager two-queue GPU schedule (bit-identical steps)
        tr.time_collectives = dist is not None
        host_s = [0.0]

        def one_step():
            t = time.perf_counter()
            tr.train_step(x, y)
            host_s[0] += time.perf_counter() - t
        for _ in range(warmup):
            one_step()
        host_s[0] = 0.0
        tr.__dict__["comm_times"] = []
        el = timed(one_step, steps, 0)
        host_loop_ms = max_over_ranks(host_s[0]) / steps * 1e3
        # the enqueue cost proper: one step issued into EMPTY queues (in the loop above a host that runs ahead of the GPU is
        # throttled by the full queue, so the in-loop figure tends to the GPU step time whatever the host costs)
        solo = []
        for _ in range(5):
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            tr.train_step(x, y)
            solo.append(time.perf_counter() - t)
        torch.cuda.synchronize(dev)
        host_ms = max_over_ranks(sorted(solo)[len(solo) // 2]) * 1e3
        loss = float(tr.last_loss.item())
        native_plan = bool(replay and isinstance(tr._plan, _lib.NativePlan))
        n_launches = None
        if replay:
            n_launches = sum(1 for fn, _ in tr._plan.calls if getattr(fn, "argtypes", None) is not None)     # C-ABI calls of one step
        # whole-job rate: every rank's images over the slowest rank's time (dist_utils.whole_job_rate: SUM of units, MAX of time;
        # `timed` has already taken the MAX, so the second reduction is the identity)
        job_rate, job_images, _ = du.whole_job_rate(TB * steps, el, device=dev if dist is not None else None)
        assert world == 1 or abs(job_images - world * TB * steps) < 0.5, "ranks disagree on their batch"
        train = {"img_per_s": job_rate, "ms_per_step": el / steps * 1e3, "loss_after": loss,
                 "launch": (("launch-plan replay, " + ("native executor (pp_plan_replay: one foreign call per step)" if native_plan else "python list"))
                            if replay else "eager") + ", small weight gradients on a second stream",
                 # host time spent inside train_step() per step (enqueue only, nothing synchronises): a host slower than the
                 # GPU step shows up HERE, not as an unexplained scaling loss
                 "host_enqueue_ms_per_step": host_ms, "host_in_loop_ms_per_step": host_loop_ms, "replay": bool(replay),
                 "replay_reason": replay_why, "launches_per_step": n_launches,
                 "host_cores_per_rank": round(cores_per_rank, 1),
                 "grad_bytes_allreduced_per_step": tr.n * 4 if world > 1 else 0}
        if headline:
            was_replay = tr._plan is not None
            if was_replay:
                tr.disable_replay()
            cm = train_step_cost_model(tr, x, y)
            if was_replay:
                tr.enable_replay(x, y, warmup=1)
            floor = max(cm["mfma_floor_ms"], cm["hbm_floor_ms"])
            cm["floor_ms"] = floor
            cm["frac"] = round(floor / train["ms_per_step"], 4)
            cm["frac_if_floors_add"] = round((cm["mfma_floor_ms"] + cm["hbm_floor_ms"]) / train["ms_per_step"], 4)
            cm["achieved_TF_fp32_equivalent"] = round(cm["flops_per_step"] / (train["ms_per_step"] * 1e-3) / 1e12, 2)
            train["roofline"] = cm
        if headline and not replay and dist is None:
            # the same step through the recorded launch list (native executor, csrc/plan.hip): what the host pays then, and what the GPU
            # step costs - reported beside the eager figures above, which stay the headline while they are the faster ones
            tr.enable_replay(x, y, warmup=1)
            for _ in range(5):
                tr.train_step(x, y)
            el_r = timed(lambda: tr.train_step(x, y), steps, 0)
            solo = []
            for _ in range(5):
                torch.cuda.synchronize(dev)
                t = time.perf_counter()
                tr.train_step(x, y)
                solo.append(time.perf_counter() - t)
            torch.cuda.synchronize(dev)
            train["native_replay"] = {"ms_per_step": round(el_r / steps * 1e3, 4), "img_per_s": round(world * TB * steps / el_r, 2),
                                      "host_enqueue_ms_per_step": round(sorted(solo)[2] * 1e3, 4),
                                      "c_abi_calls_per_step": sum(1 for fn, _ in tr._plan.calls if getattr(fn, "argtypes", None) is not None),
                                      "executor": "pp_plan_replay (one foreign call per step)" if isinstance(tr._plan, _lib.NativePlan) else "python list"}
            tr.disable_replay()
        return train, tr, model

    train = None
    if a.mode in ("both", "train"):
        from pixelpick_amd import engine as E
        TB = a.train_batch
        train, tr, model = train_leg(a.network, a.steps, a.warmup, H, W, C, headline=True)
        if dist is not None:
            # what the communicator really is, and what the gradient exchange costs on its own (both buckets back to back,
            # nothing else running): the overlapped step hides most of the first bucket under the encoder backward
            ar = []
            for _ in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                barrier()
                e0.record()
                dist.all_reduce(tr.flat_g[tr.n_split:])
                dist.all_reduce(tr.flat_g[:tr.n_split])
                e1.record()
                torch.cuda.synchronize(dev)
                ar.append(e0.elapsed_time(e1))
            ar_ms = max_over_ranks(sorted(ar[1:])[len(ar[1:]) // 2])
            in_step = {}
            for tag, e0, e1 in tr.__dict__.get("comm_times", []):
                in_step.setdefault(tag, []).append(e0.elapsed_time(e1) * 1e3)
            tr.time_collectives = False
            # --- what the first run on a real multi-GPU node needs in order to be trusted (nobody has had one yet) ---
            # (1) which device every rank really sits on; ranks sharing a device (the one-GPU plumbing runs over gloo) are flagged
            import socket
            rank_devs = [None] * world
            dist.all_gather_object(rank_devs, {"rank": rank, "local_rank": local_rank, "device": dev_index, "host": socket.gethostname(),
                                               "pci_bus_id": getattr(torch.cuda.get_device_properties(dev_index), "pci_bus_id", None),
                                               "uuid": str(getattr(torch.cuda.get_device_properties(dev_index), "uuid", ""))})
            shared = len({(d["host"], d["device"]) for d in rank_devs}) < world
            # (2) the channels the communicator opened (each keeps a block resident on a CU during a collective) beside the CUs the
            # spin-waiting launches leave free for them
            channels = du.rccl_channels_from_log(rccl_log) if rccl_log else None
            # (3) the N = 1 figure of THIS invocation: rank 0 alone on its device, its own one-rank group (no collective), the other
            # ranks idle at the barrier - what SCALE's N = 1 point and BENCH's headline should both agree with on this box
            solo_groups = [dist.new_group([r]) for r in range(world)]          # (every rank creates every group)
            n1 = None
            if rank == 0:
                from pixelpick_amd.trainer import FlatTrainer
                from pixelpick_amd.utils.utils import get_model
                torch.manual_seed(0)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    m1 = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name=a.network, weight_type="random",
                                             n_layers=50, use_softmax=True, use_dilated_resnet=True, width_multiplier=1.0)).to(dev).train()
                tr1 = FlatTrainer(m1, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=2e-4, ignore_index=C, process_group=solo_groups[0])
                assert tr1.world == 1 and not tr1.collectives
                x1, y1 = synth_train_batch(TB, C, H, W, a.n_labelled, dev, 1)
                if train["replay"]:
                    tr1.enable_replay(x1, y1, warmup=1)
                for _ in range(max(a.warmup, 3)):
                    tr1.train_step(x1, y1)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(a.steps):
                    tr1.train_step(x1, y1)
                torch.cuda.synchronize(dev)
                el1 = time.perf_counter() - t1
                n1 = {"img_per_s": round(TB * a.steps / el1, 2), "ms_per_step": round(el1 / a.steps * 1e3, 4), "replay": bool(train["replay"]),
                      "what": "rank 0 alone (one-rank group, no all-reduce) while the other ranks wait: the N = 1 point of this very run"}
                tr1.close()
                del tr1, m1, x1, y1
                torch.cuda.empty_cache()
            barrier()
            line["distributed"] = {"backend": dist.get_backend(), "nranks": dist.get_world_size(), "devices_visible": torch.cuda.device_count(),
                                   "rank_devices": rank_devs, "shared_device": bool(shared),
                                   "comm_cu_reserve": int(L.pp_get_comm_cu_reserve()), "rccl_channels": channels,
                                   "rccl_channels_source": ("NCCL_DEBUG=INFO log of this run (the 'coll channels' line of communicator init)" if channels is not None
                                                            else ("not an RCCL communicator" if dist.get_backend() != "nccl" else "no channel line in the RCCL log")),
                                   "rccl_env": {k: os.environ[k] for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "PIXELPICK_COMM_CU_RESERVE") if k in os.environ},
                                   "n1_same_invocation": n1,
                                   "scaling_vs_n1_same_invocation": (round(train["img_per_s"] / (world * n1["img_per_s"]), 4) if n1 else None),
                                   "allreduce_in_step_us": {k: round(sum(v) / len(v), 1) for k, v in in_step.items()},
                                   "allreduce_bytes_per_step": tr.n * 4,
                                   "buckets": ([int((tr.n - tr.n_split) * 4), int((tr.n_split - tr.n_mid) * 4), int(tr.n_mid * 4)] if tr.n_mid
                                               else [int((tr.n - tr.n_split) * 4), int(tr.n_split * 4)]),
                                   "allreduce_alone_ms": round(ar_ms, 4),
                                   "allreduce_alone_GBps_per_rank": round(tr.n * 4 / (ar_ms * 1e-3) / 1e9, 1),
                                   "overlapped": bool(__import__("pixelpick_amd.trainer", fromlist=["x"]).OVERLAP_ALLREDUCE)}

        # the depthwise layers of MobileNetV2 against the HBM roofline (north_star: "bandwidth-bound and justified against the HBM roofline";
        # SURVEY 8(d)): every depthwise call of the RECORDED step re-issued on its own with the step's arguments (pixelpick_amd/profiling.py;
        # the per-layer table with cold timings and copy yardsticks is tools/dw_bench.py -> profiles/r06_dw_layers.txt)
        if world == 1 and a.network == "deeplab" and not a.no_other_configs:
            from pixelpick_amd import profiling
            was_replay = tr._plan is not None
            if not was_replay:
                tr.enable_replay(*synth_train_batch(TB, C, H, W, a.n_labelled, dev, 1), warmup=1)
            rows_dw = profiling.depthwise_table(tr._plan, iters=10, with_cold=False, yardsticks=False)
            if not was_replay:
                tr.disable_replay()
            line["roofline_hbm_depthwise"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "what": "mobilenet_v2.py:38,52 - the 17 depthwise 3x3 layers of one train step (B = %d, %dx%d), each launch timed alone with the "
                                                      "step's own arguments, warm; bytes = every tensor once" % (TB, H, W),
                                              **profiling.depthwise_summary(rows_dw)}
        # dominant train kernel vs the fp32 MFMA roofline: SegmentHead conv 3x3 304->256 on [TB,64,128] (decoders.py:107)
        Hq, Wq = H // 4, W // 4
        xa = torch.randn((TB, Hq, Wq, 304), device=dev)
        wa = torch.randn((3, 3, 304, 256), device=dev) * 0.02
        ya = torch.empty((TB, Hq, Wq, 256), device=dev)
        nrep = max(a.steps, 10)
        # the op as the train step runs it: operand splits (x3_split_kernel, x3_split_w_kernel) + conv_x3_kernel - the fp32
        # convolution on the bf16 matrix pipe (every operand = hi + mid + lo bf16, six MFMAs per product, fp32 accumulate;
        # error against float64 = that of the fp32-MFMA kernel, tests/test_conv_x3_gpu.py) - and, for reference, the fp32-MFMA
        # kernel it replaced (test build: pp_debug_set_x3(0))
        wsx = torch.empty(max(int(L.pp_conv2d_fwd_workspace_bytes(TB, Hq, Wq, 304, 256, 3, 3, 1, 1, 1)), 256), dtype=torch.uint8, device=dev)

        def conv_once():
            rc = L.pp_conv2d_fwd(xa.data_ptr(), 304, TB, Hq, Wq, 304, wa.data_ptr(), None, 3, 3, 1, 1, 1, ya.data_ptr(), 256, 256,
                                 wsx.data_ptr(), wsx.numel(), stream)
            _lib.check(rc, "pp_conv2d_fwd")
        flops = 2.0 * TB * Hq * Wq * 256 * 9 * 304
        res = {}
        for tag, mode in (("bf16x3", 1), ("fp32_mfma", 0)):
            if not KNOBS and mode == 0:      # the fp32-MFMA kernel on this layer is a planner switch: test build only (--knobs-build)
                res[tag] = None
                continue
            if KNOBS:
                L.pp_debug_set_x3(mode)
            evc = HipEvents(nrep)
            timed(conv_once, nrep, 12, evc)      # warm: the first launches after the train loop read ~10 % low (clock ramp)
            cms = evc.elapsed_ms()
            evc.destroy()
            res[tag] = sum(cms) / len(cms)
        if KNOBS:
            L.pp_debug_set_x3(1)
        cavg = res["bf16x3"]
        # The kernel runs on the bf16 matrix pipe and spends six bf16 MFMAs per fp32 product: ITS roof is the dense bf16 peak / 6.
        # (Rounds 2-3 divided by the fp32 MFMA peak and printed a "fraction" above 1 - the wrong roof; that ratio stays below as a
        # secondary, clearly named key.)
        X3_PEAK_TF = MFMA_BF16_PEAK_TF / 6.0
        ach = flops / (cavg * 1e-3) / 1e12
        line["roofline_mfma"] = {"bound": "mfma", "kernel": "x3_split_kernel + x3_split_w_kernel + conv_x3_kernel<256,128>, SegmentHead 3x3 304->256 fwd",
                                 "arithmetic": "fp32 operands split exactly into 3 bf16 planes, 6 x v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate",
                                 "achieved": round(ach, 2), "peak": round(X3_PEAK_TF, 1), "unit": "TFLOP/s (fp32-equivalent: 2*M*N*K / time)",
                                 "frac": round(ach / X3_PEAK_TF, 4),
                                 "peak_note": f"dense bf16 MFMA peak {MFMA_BF16_PEAK_TF} TF / 6 MFMAs per fp32 product (MI355X_MICROARCH.md)",
                                 "bf16_pipe_achieved_TF": round(6 * ach, 1), "bf16_pipe_peak_TF": MFMA_BF16_PEAK_TF,
                                 "vs_fp32_mfma_peak": round(ach / MFMA_F32_PEAK_TF, 4),
                                 "fp32_mfma_kernel": ({"kernel": "conv_igemm_dma_kernel<128,128>", "kernel_ms_avg": round(res["fp32_mfma"], 4),
                                                       "achieved": round(flops / (res["fp32_mfma"] * 1e-3) / 1e12, 2), "peak": MFMA_F32_PEAK_TF,
                                                       "frac": round(flops / (res["fp32_mfma"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4)}
                                                      if res["fp32_mfma"] else "A/B against the fp32-MFMA kernel is a planner switch of the test build: python bench.py --knobs-build"),
                                 "traffic": None, "algorithmic_flops_per_launch": flops, "kernel_ms_avg": round(cavg, 4)}
        # conv_x3_kernel ALONE: both operands' planes held by the caller, as in the train step (the activation planes are shared with the
        # weight gradient, the weight planes are split at begin_step on the second queue: pp_conv2d_fwd_pre2)
        xpl = torch.empty(int(L.pp_x3_planes_bytes(TB * Hq * Wq, 304)), dtype=torch.uint8, device=dev)
        wpl = torch.empty(int(L.pp_x3_weight_planes_bytes(9, 304, 256, 1)), dtype=torch.uint8, device=dev)
        _lib.check(L.pp_x3_split(xa.data_ptr(), 304, TB * Hq * Wq, 304, xpl.data_ptr(), xpl.numel(), stream), "pp_x3_split")
        _lib.check(L.pp_x3_split_weights(wa.data_ptr(), 9, 304, 256, 1, wpl.data_ptr(), wpl.numel(), stream), "pp_x3_split_weights")

        def conv_only():
            _lib.check(L.pp_conv2d_fwd_pre2(xa.data_ptr(), 304, TB, Hq, Wq, 304, wa.data_ptr(), None, 3, 3, 1, 1, 1, ya.data_ptr(), 256, 256,
                                            wsx.data_ptr(), wsx.numel(), xpl.data_ptr(), wpl.data_ptr(), stream), "pp_conv2d_fwd_pre2")
        evk = HipEvents(nrep)
        timed(conv_only, nrep, 12, evk)
        kms = evk.elapsed_ms()
        evk.destroy()
        kavg = sum(kms) / len(kms)
        ach_k = flops / (kavg * 1e-3) / 1e12
        line["roofline_mfma"]["kernel_only"] = {"what": "conv_x3_kernel<256,128> with both operands' planes held by the caller (pp_conv2d_fwd_pre2)",
                                                "kernel_ms_avg": round(kavg, 4), "achieved": round(ach_k, 2), "frac": round(ach_k / X3_PEAK_TF, 4)}
        # What the matrix pipe SUSTAINS on this box under conv_x3_kernel's own MFMA stream and nothing else (pp_yardstick_mfma_stream:
        # register-resident fragments, launches of the convolution's length): the spec peak assumes 2.4 GHz, and a chip-wide bf16 MFMA
        # load on operands with random mantissas is power-limited well below it (profiles/r05_conv_x3_power.txt) - zeros are not.
        sink = torch.zeros(64, device=dev)
        mf = {}
        for kind, tag in ((0, "zero_operands"), (2, "random_operands")):
            it0 = 160
            def stream_once():
                _lib.check(L.pp_yardstick_mfma_stream(kind, it0, sink.data_ptr(), stream), "pp_yardstick_mfma_stream")
            evs = HipEvents(nrep)
            timed(stream_once, nrep, 12, evs)
            sms = evs.elapsed_ms()
            evs.destroy()
            sm = sum(sms) / len(sms)
            n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
            mf[tag] = {"bf16_TF": round(n_cu * 8 * it0 * 24 * 2.0 * 32 * 32 * 16 / (sm * 1e-3) / 1e12, 1), "kernel_ms_avg": round(sm, 4)}
        sus = mf["random_operands"]["bf16_TF"] / 6.0
        line["roofline_mfma"]["sustained"] = {
            "what": "pp_yardstick_mfma_stream: conv_x3_kernel's six-term MFMA sequence from registers only (no LDS, no DMA, no barrier), 8 waves/CU, "
                    "a launch of the convolution's length - the rate the power limit leaves on this box",
            **mf, "fp32_equivalent_peak_random_operands_TF": round(sus, 1),
            "frac_of_spec_peak": round(mf["random_operands"]["bf16_TF"] / MFMA_BF16_PEAK_TF, 4),
            "conv_x3_frac_of_sustained": round(ach / sus, 4), "conv_x3_kernel_only_frac_of_sustained": round(ach_k / sus, 4)}
        # SURVEY 8(d) graded 1x1 shapes at the BASELINE batch: op time (split-K launch + its reduce where the plan
        # splits) from HIP events; ceiling = min(MFMA peak, arithmetic intensity x HBM peak) for ONE pass over x, w, y.
        graded = [("ASPP fuse 1280->256 @16x32 (aspp.py:73-75)", 16, 32, 1280, 256),
                  ("MNv2 expand 160->960 @18x34 (mobilenet_v2.py:42)", 18, 34, 160, 960),
                  ("MNv2 project 960->160 @16x32 (mobilenet_v2.py:56)", 16, 32, 960, 160),
                  ("MNv2 project 960->320 @16x32 (mobilenet_v2.py:56)", 16, 32, 960, 320),
                  ("R50 Bottleneck 256->1024 @32x64 (resnet_models.py:66)", 32, 64, 256, 1024),
                  ("R50 Bottleneck 1024->256 @32x64 (resnet_models.py:60)", 32, 64, 1024, 256),
                  ("R50 Bottleneck 2048->512 @32x64 (resnet_models.py:60)", 32, 64, 2048, 512),
                  ("R50 Bottleneck 64->256 @64x128 (resnet_models.py:66)", 64, 128, 64, 256)]
        def time_1x1(bb, h1, w1, ci, co):
            xg = torch.randn((bb, h1, w1, ci), device=dev)
            wg = torch.randn((1, 1, ci, co), device=dev) * 0.05
            yg = torch.empty((bb, h1, w1, co), device=dev)
            wsb = int(L.pp_conv2d_fwd_workspace_bytes(bb, h1, w1, ci, co, 1, 1, 1, 0, 1))
            wsg = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
            evg = HipEvents(10)

            def one():
                rc = L.pp_conv2d_fwd(xg.data_ptr(), ci, bb, h1, w1, ci, wg.data_ptr(), None, 1, 1, 1, 0, 1, yg.data_ptr(), co, co,
                                     wsg.data_ptr() if wsb else None, wsb, stream)
                _lib.check(rc, "pp_conv2d_fwd")
            timed(one, 10, 3, evg)
            ms1 = sum(evg.elapsed_ms()) / 10
            evg.destroy()
            return ms1, bool(wsb)

        def time_lib_sgemm(bb, h1, w1, ci, co):
            """Yardstick only (never on the product path): the vendor library's fp32 GEMM on the same M x K x N, torch.matmul
            into a preallocated output, timed with torch events on the stream it runs on."""
            m1 = bb * h1 * w1
            xg = torch.randn((m1, ci), device=dev)
            wg = torch.randn((ci, co), device=dev) * 0.05
            yg = torch.empty((m1, co), device=dev)
            for _ in range(5):
                torch.matmul(xg, wg, out=yg)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            torch.cuda.synchronize(dev)
            for e0, e1 in evs:
                e0.record()
                torch.matmul(xg, wg, out=yg)
                e1.record()
            torch.cuda.synchronize(dev)
            return sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs)

        rows1 = []
        BIG = 16 * TB          # the same layers at 16x the rows (e.g. a 64-image acquisition/validation forward): the
        for name, h1, w1, ci, co in graded:      # kernel's rate once the grid fills the chip
            ms1, split = time_1x1(TB, h1, w1, ci, co)
            msb, _ = time_1x1(BIG, h1, w1, ci, co)
            m1 = TB * h1 * w1
            fl = 2.0 * m1 * ci * co
            by = 4.0 * (m1 * ci + ci * co + m1 * co)
            ceil_tf = min(MFMA_F32_PEAK_TF, fl / by * HBM_PEAK_GBS / 1e3)
            tfb = 16 * fl / (msb * 1e-3) / 1e12
            lib_ms = time_lib_sgemm(TB, h1, w1, ci, co)
            rows1.append({"shape": name, "rows": m1, "split_k": split, "us": round(ms1 * 1e3, 2),
                          "library_sgemm_us": round(lib_ms * 1e3, 2), "library_sgemm_tf": round(fl / (lib_ms * 1e-3) / 1e12, 2),
                          "vs_library": round(lib_ms / ms1, 3),
                          "achieved": round(fl / (ms1 * 1e-3) / 1e12, 2), "ceiling": round(ceil_tf, 1),
                          "frac_of_mfma_peak": round(fl / (ms1 * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
                          "achieved_at_16x_rows": round(tfb, 2), "frac_at_16x_rows": round(tfb / MFMA_F32_PEAK_TF, 4)})
        line["roofline_mfma_1x1"] = {"unit": "TFLOP/s", "peak": MFMA_F32_PEAK_TF, "batch": TB, "batch_16x": BIG,
                                     "yardstick": "library_sgemm_* = torch.matmul fp32 (vendor BLAS) on the same M x K x N, measurement only",
                                     "shapes": rows1}
        tr.disable_replay()
        del tr, model, xa, wa, ya
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------------------------ acquisition
    def acq_leg(B, Ca, Ha, Wa, ka, strategy, layout, steps, warmup, headline):
        """K passes of pp_acq_score_topk over B resident images of [Ca, Ha, Wa] logits -> (record, roofline of acq_kernel).
        headline: also the counter-traffic record, and the selection straight from 1/4-resolution logits."""
        gen = torch.Generator(device=dev).manual_seed(100 + rank)
        logits = torch.randn((B, Ca, Ha, Wa), device=dev, generator=gen) * 3        # resident in HBM before timing
        if layout == "nhwc":
            logits = logits.contiguous(memory_format=torch.channels_last)
        excl = (torch.rand((B, Ha, Wa), device=dev, generator=gen) < 0.05).to(torch.uint8)
        idx = torch.empty((B, ka), dtype=torch.int32, device=dev)
        val = torch.empty((B, ka), dtype=torch.float32, device=dev)
        ws = torch.empty(max(L.pp_acq_workspace_bytes(B, Ca, Ha, Wa, ka), 256), dtype=torch.uint8, device=dev)
        sB, sC, sH, sW = logits.stride()
        sid = acq.STRATEGY_ID[strategy] | exact_flag[0]

        def acq_step():
            rc = L.pp_acq_score_topk(logits.data_ptr(), B, Ca, Ha, Wa, sB, sC, sH, sW, excl.data_ptr(), sid, ka,
                                     idx.data_ptr(), val.data_ptr(), None, ws.data_ptr(), ws.numel(), stream)
            _lib.check(rc, "pp_acq_score_topk")
        ev = HipEvents(steps)
        el = timed(acq_step, steps, warmup, ev)
        kms = ev.elapsed_ms()
        ev.destroy()
        alg_bytes = B * Ha * Wa * (4 * Ca + 1)               # per launch of acq_kernel on ONE GPU
        kavg = sum(kms) / len(kms)
        achieved = alg_bytes / (kavg * 1e-3) / 1e9
        acqr = {"value": round(du.whole_job_rate(B * Ha * Wa * steps, el, device=dev if dist is not None else None)[0] / 1e6, 1), "unit": "Mpixels/s",
                "ms_per_step": round(el / steps * 1e3, 4), "images_per_launch_per_gpu": B, "k": ka,
                "strategy": strategy, "layout": layout}
        # HBM traffic per launch: a counter pass cannot run inside this process, so it comes from the record a committed
        # script writes (tools/measure_acq_traffic.py: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled as
        # MI355X_MICROARCH.md prescribes for gfx950).  The record carries the hash of the kernel source it measured; if this
        # build's source differs, or the configuration is another one, the line says null instead of a stale number.
        traffic, traffic_src = None, None
        rec_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "acq_traffic.json")
        if headline and os.path.exists(rec_path):
            rec = json.load(open(rec_path))
            cfg = rec.get("config", {})
            same_cfg = (cfg.get("B"), cfg.get("C"), cfg.get("H"), cfg.get("W"), cfg.get("k"), cfg.get("strategy"), cfg.get("layout")) == \
                       (B, Ca, Ha, Wa, ka, strategy, layout)
            if same_cfg and rec.get("acq_source_sha256") == _acq_source_hash():
                traffic = rec["traffic_bytes_per_launch"]
                traffic_src = f"profiles/acq_traffic.json ({rec['kernel'].split('(')[0]}, measured {rec['measured_unix']})"
            elif same_cfg:
                traffic_src = "profiles/acq_traffic.json is for another build of csrc/acq.hip: re-run tools/measure_acq_traffic.py"
        # yardstick, measurement only: a kernel that does nothing but read the same logits buffer (one wave per SIMD, eight 16-byte
        # loads in flight per lane - the fastest form tools/probe/hbm_rw.hip found): what "bandwidth-bound" can reach on this box
        yard = None
        if layout == "nchw":
            sink = torch.zeros(4, device=dev)
            nbytes = logits.numel() * 4
            for _ in range(3):
                _lib.check(L.pp_yardstick_stream_read(logits.data_ptr(), nbytes, 256, sink.data_ptr(), stream), "pp_yardstick_stream_read")
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            torch.cuda.synchronize(dev)
            for e0, e1 in evs:
                e0.record()
                _lib.check(L.pp_yardstick_stream_read(logits.data_ptr(), nbytes, 256, sink.data_ptr(), stream), "pp_yardstick_stream_read")
                e1.record()
            torch.cuda.synchronize(dev)
            yms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)[len(evs) // 2]
            yard_plain = nbytes / (yms * 1e-3) / 1e9
            # the same with non-temporal loads (blocks < 0) - what the scorers use since round 5; the yardstick is the better of the two
            for _ in range(3):
                _lib.check(L.pp_yardstick_stream_read(logits.data_ptr(), nbytes, -256, sink.data_ptr(), stream), "pp_yardstick_stream_read")
            torch.cuda.synchronize(dev)
            for e0, e1 in evs:
                e0.record()
                _lib.check(L.pp_yardstick_stream_read(logits.data_ptr(), nbytes, -256, sink.data_ptr(), stream), "pp_yardstick_stream_read")
                e1.record()
            torch.cuda.synchronize(dev)
            yms_nt = sorted(e0.elapsed_time(e1) for e0, e1 in evs)[len(evs) // 2]
            yard_nt = nbytes / (yms_nt * 1e-3) / 1e9
            yard = max(yard_plain, yard_nt)
        roof = {"bound": "hbm", "kernel": "acq_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_avg": round(kavg, 4),
                "kernel_ms_min": round(min(kms), 4),
                "read_only_yardstick": ({"GB/s": round(yard, 1), "frac_of_peak": round(yard / HBM_PEAK_GBS, 4),
                                         "acq_kernel_vs_yardstick": round(achieved / yard, 4),
                                         "ordinary_loads_GB/s": round(yard_plain, 1), "non_temporal_loads_GB/s": round(yard_nt, 1),
                                         "what": "pp_yardstick_stream_read: a kernel that only reads the same logits buffer (the better of ordinary and non-temporal loads)"}
                                        if yard else None)}

        if ka > 48:
            # large k (round 6): no score map any more - the scorer writes only the (key, index) words of the pixels beyond a sampled
            # threshold key (~1 B / pixel, not counted as algorithmic), the list select picks from them; the op-level fraction is the
            # figure that includes the sample launch and the select
            roof["selection"] = "sampled threshold -> candidate emission in the scorer -> list select (exact fallback for flagged images)"
            roof["op_frac"] = round(alg_bytes / (el / steps) / 1e9 / HBM_PEAK_GBS, 4)
        # the same selection straight from the classifier output at 1/4 resolution (what DeepLab's
        # head writes before deeplab.py:55-56), interpolated on the fly - the production path of QuerySelector for DeepLab
        if headline and layout == "nchw" and Ha % 4 == 0 and Wa % 4 == 0:
            low = torch.randn((B, Ha // 4, Wa // 4, Ca), device=dev, generator=gen) * 3
            ws2 = torch.empty(max(L.pp_acq_lowres_workspace_bytes(B, Ca, Ha, Wa, ka), 256), dtype=torch.uint8, device=dev)

            def lowres_step():
                rc = L.pp_acq_lowres_score_topk(low.data_ptr(), Ca, B, Ca, Ha // 4, Wa // 4, Ha, Wa, 1, Ha, Wa, excl.data_ptr(), sid, ka,
                                                idx.data_ptr(), val.data_ptr(), None, ws2.data_ptr(), ws2.numel(), stream)
                _lib.check(rc, "pp_acq_lowres_score_topk")
            el2 = timed(lowres_step, steps, warmup)
            acqr["from_lowres_logits"] = {"value": round(world * B * Ha * Wa * steps / el2 / 1e6, 1), "unit": "Mpixels/s",
                                          "ms_per_step": round(el2 / steps * 1e3, 4), "kernel": "acq_lowres_kernel",
                                          "input": f"[{B},{Ha // 4},{Wa // 4},{Ca}] channels-last logits, bilinear x4 align_corners folded in",
                                          "bound": "valu (interpolation + softmax arithmetic; 16x less input than the full-size logits)"}
            del low, ws2
        del logits, excl, idx, val, ws
        torch.cuda.empty_cache()
        return acqr, roof

    acqr = None
    if a.mode in ("both", "acq"):
        acqr, line["roofline"] = acq_leg(a.batch, C, H, W, k, a.strategy, a.layout, a.steps, a.warmup, True)
        if not a.exact_formula and world == 1 and not a.no_other_configs:
            # the reference's operation order as the scorer (SURVEY 7.4 wanted it selectable; the default is the algebraic form):
            # same launch, same traffic, more VALU work per pixel
            exact_flag[0] = 0x100
            rx, roofx = acq_leg(a.batch, C, H, W, k, a.strategy, a.layout, max(5, min(a.steps, 10)), 3, False)
            exact_flag[0] = 0
            acqr["exact_formula"] = {"value": rx["value"], "unit": rx["unit"], "ms_per_step": rx["ms_per_step"],
                                     "kernel_ms_avg": roofx["kernel_ms_avg"], "frac_of_hbm_peak": roofx["frac"],
                                     "what": "strategy | PP_ACQ_REFERENCE_ORDER: p = exp(x - m) / S, sum(-p log p) in query.py:190,230's order (a per-call flag of the C ABI)"}

    # ------------------------------------------------------------------------------------ the other BASELINE configurations
    # Bounded sub-records in the SAME run (N = 1 only; the scaling runs stay short): every acquisition shape / strategy the
    # BASELINE.json configs name plus the reference's default top-5 % mode (args.py:25), each with its own roofline of acq_kernel,
    # and the ResNet50 train steps of configs[2] (the reference's ResNet50 model FPNSeg, and DeepLabv3+-ResNet50 as named).
    if a.mode == "both" and world == 1 and not a.no_other_configs and a.network == "deeplab" and (C, H, W) == (19, 256, 512):
        others = []
        oc_steps, oc_warm = max(5, min(a.steps, 20)), 5
        for name, (Bo, Co, Ho, Wo, ko, so) in (
                ("configs[0] CamVid 360x480, C=11, entropy top-20", (128, 11, 360, 480, 20, "entropy")),
                ("configs[1] Cityscapes 256x512, C=19, entropy, reference default top-5 % (k = 6553, args.py:25)", (256, 19, 256, 512, 6553, "entropy")),
                ("configs[3] VOC 320x320 crops, C=21, margin top-20", (256, 21, 320, 320, 20, "margin_sampling")),
                ("configs[4] Cityscapes 1024x2048, C=19, least-confidence top-20", (8, 19, 1024, 2048, 20, "least_confidence"))):
            r, roof = acq_leg(Bo, Co, Ho, Wo, ko, so, "nchw", oc_steps, oc_warm, False)
            roof.pop("traffic_source", None)
            others.append({"config": name, "leg": "acquisition", "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                           "images_per_launch": Bo, "roofline": roof})
        for name, net in (("configs[2] Cityscapes 256x512, ResNet50 model of the reference (FPNSeg), train step", "FPN"),
                          ("configs[2] as named: DeepLabv3+-ResNet50 (assembled from the reference's parts), train step", "deeplab_r50")):
            t2, tr2, m2 = train_leg(net, max(5, min(a.steps, 10)), 4, H, W, C)
            tr2.disable_replay()
            del tr2, m2
            torch.cuda.empty_cache()
            others.append({"config": name, "leg": "train", "value": round(t2["img_per_s"], 2), "unit": "images/s",
                           "ms_per_step": round(t2["ms_per_step"], 4), "per_gpu_batch": a.train_batch, "replay": t2["replay"],
                           "host_enqueue_ms_per_step": round(t2["host_enqueue_ms_per_step"], 3)})
        # the acquisition ROUND of configs[2-4] end to end (QuerySelector.__call__: eval forward of the ResNet50 model, fused
        # low-resolution tail, picks, codec, statistics; tools/query_bench.py) beside the reference-order path of the same build
        import importlib.util
        spec = importlib.util.spec_from_file_location("pp_query_bench", os.path.join(ROOT, "tools", "query_bench.py"))
        qb = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(qb)
        for label, net, Cq, Hq, Wq, stq, dsn, ign, nimg, bs, nu in qb.CONFIGS:
            mq = qb.build_model(net, Cq)
            rf = qb.round_rate(net, Cq, Hq, Wq, stq, nimg, bs, dsn, ign, fused=True, model=mq, n_unique=nu)
            torch.cuda.empty_cache()
            ru = qb.round_rate(net, Cq, Hq, Wq, stq, nimg, bs, dsn, ign, fused=False, model=mq, n_unique=nu)
            torch.cuda.empty_cache()
            tt = qb.tail_times(mq, Cq, (Hq + 7) // 8 * 8, (Wq + 7) // 8 * 8, stq, batch=bs)
            del mq
            torch.cuda.empty_cache()
            others.append({"config": label, "leg": "acquisition_round", "value": round(rf["images_per_s"], 2), "unit": "images/s",
                           "ms_per_image": round(rf["ms_per_image"], 3), "images": nimg, "images_per_forward": bs, "k": 20,
                           "peak_device_gib": round(rf["peak_gib"], 2),
                           "reference_order_path": {"value": round(ru["images_per_s"], 2), "ms_per_image": round(ru["ms_per_image"], 3),
                                                    "peak_device_gib": round(ru["peak_gib"], 2)},
                           "tail_only_gpu_ms": {k2: round(v, 3) for k2, v in tt.items()},
                           "tail_speedup": round(max(tt["tail_reference_order_ms"], 0.0) / max(tt["tail_fused_ms"], 1e-3), 1)})
        line["other_configs"] = others

    if rank == 0:
        head = {"n_gpus": world, "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic", "library_build": library_build,
                "dtype_note": "fp32 tensors and fp32 accumulation throughout; the MFMA-bound 3x3 / large 1x1 convolutions split each fp32 operand "
                              "exactly into three bf16 terms and run six bf16 MFMAs per product (error vs float64 equal to the fp32-MFMA "
                              "kernels', tests/test_conv_x3_gpu.py; the test build's pp_debug_set_x3(0) selects the fp32-MFMA kernels)"}
        net_desc = {"deeplab": "BASELINE configs[1]: Cityscapes {}x{}, C={}, DeepLabv3+-MobileNetV2",
                    "FPN": "BASELINE configs[2] (per GPU): Cityscapes {}x{}, C={}, ResNet50 model of the reference (FPNSeg)",
                    "deeplab_r50": "BASELINE configs[2] as named (per GPU): Cityscapes {}x{}, C={}, DeepLabv3+-ResNet50 assembled from "
                                   "the reference's parts (dilated ResNet50 + ASPP rates 12/24/36 + SegmentHead; not a reference model)"
                    }[a.network].format(H, W, C)
        wl = (f"{net_desc}; train step per-GPU batch "
              f"{a.train_batch} with {a.n_labelled} labelled px/img + Adam; acquisition {a.strategy} top-k={k}, "
              f"B={a.batch} images/launch/GPU, {a.layout.upper()} fp32 logits")
        if train is not None:
            out = {"metric": "train_step_throughput", "value": round(train["img_per_s"], 2), "unit": "images/s",
                   "ms_per_step": round(train["ms_per_step"], 4)}
        else:
            out = {"metric": "acquisition_throughput", "value": acqr["value"], "unit": "Mpixels/s", "ms_per_step": acqr["ms_per_step"]}
        out.update(head)
        out["config"] = {"workload": wl, "global_batch": world * a.train_batch,
                         "sharding": f"images over {world} rank(s); train: the flat gradient is all-reduced per step in three pieces (everything "
                                     f"behind the encoder and then the late encoder blocks under the encoder backward, the early encoder after it); "
                                     f"acquisition: no collective"}
        if train is not None:
            out["train"] = {k2: (round(v, 4) if isinstance(v, float) else v) for k2, v in train.items()}
        if acqr is not None:
            out["acquisition"] = acqr
        out.update(line)
        if world == 1 and not a.no_cpu_baseline:
            # torch's default of one intra-op thread per core (128 here) is the WORST setting for these ops on this
            # host (fork/join dominates: 0.96 img/s and 15 Mpix/s vs 5.7 and 146 at 16 threads, measured round 1), so the
            # port is timed at a few thread counts and the best one is the baseline; `cores` = the threads it used.
            default_threads = torch.get_num_threads()
            cands = sorted({t for t in (8, 16, 32) if t <= default_threads} | {min(default_threads, 16)})
            cb = {"kind": "port", "threads_tried": cands, "torch_default_threads": default_threads}
            if train is not None:
                best = None
                for t in cands:
                    torch.set_num_threads(t)
                    v, sample, thr = cpu_baseline_train(a.train_batch, C, H, W, a.n_labelled, budget_s=5.0)
                    if best is None or v > best[0]:
                        best = (v, sample, t)
                cb.update({"value": best[0], "unit": "images/s", "sample": best[1], "cores": best[2]})
            if acqr is not None:
                best = None
                for t in cands:
                    torch.set_num_threads(t)
                    v, sample = cpu_baseline_acq(C, H, W, k, a.strategy, budget_s=2.5)
                    if best is None or v > best[0]:
                        best = (v, sample + f", {t} threads", t)
                if train is None:
                    cb.update({"value": best[0], "unit": "Mpixels/s", "sample": best[1], "cores": best[2]})
                else:
                    cb["acquisition"] = {"value": best[0], "unit": "Mpixels/s", "sample": best[1], "cores": best[2]}
            torch.set_num_threads(default_threads)
            out["cpu_baseline"] = cb
            if acqr is not None and not a.exact_formula and not a.no_other_configs:      # (--no-other-configs: the headline launches only - the kernel-stats run)
                out["acquisition"]["index_flips_vs_oracle"] = cpu_baseline_pick_flips(dev)
            if train is not None and a.network == "deeplab":
                torch.set_num_threads(cb.get("cores", 16))
                out["train"]["free_running_gradient_deviation_vs_oracle"] = cpu_baseline_grad_deviation(C, H, W, a.train_batch, a.n_labelled, dev)
                torch.set_num_threads(default_threads)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
