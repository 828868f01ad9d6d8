#!/usr/bin/env python3
"""End-to-end acquisition round (QuerySelector.__call__: eval forward + score + top-k + codec + stats, query.py:144-221) in
images/s - for the per-image loop of query.py:159 and for the batched forward (query_batch_size), for every model the BASELINE
configs name: DeepLabv3+-MobileNetV2 (configs[1]), the reference's ResNet50 model FPNSeg (configs[2], [4]) and the assembled
DeepLabv3+-ResNet50 (configs[3]).  `round_rate()` is what bench.py's `other_configs` acquisition-round records call.

    python tools/query_bench.py                     # the DeepLab table (k = 20 and the top-5 % default, batch 1 / 4 / 16)
    python tools/query_bench.py --configs           # configs[2-4]: fused low-resolution tail vs the reference-order path
"""
import contextlib
import io
import os
import sys
import tempfile
import time
import warnings
from argparse import Namespace

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import query as ppq                      # noqa: E402
from pixelpick_amd.utils.utils import get_model             # noqa: E402


class _DS:
    """n images of one size; `n_unique` distinct tensors behind them (a 1024x2048 image is 25 MB of host memory)."""

    def __init__(self, n, C, H, W, ignore_index, n_unique=None):
        g = torch.Generator().manual_seed(0)
        u = min(n, n_unique or n)
        self.n = n
        self.xs = [torch.randn(3, H, W, generator=g) for _ in range(u)]
        self.ys = [torch.randint(0, C, (H, W), generator=g) for _ in range(u)]
        for y in self.ys:
            y[torch.rand(H, W, generator=g) < 0.03] = ignore_index
        self.queries = [np.zeros((H, W), bool) for _ in range(n)]
        self.image_sizes = [(H, W)] * n

    def label_queries(self, d, nth):
        pass


class _DL:
    def __init__(self, ds):
        self.dataset = ds

    def __iter__(self):
        ds = self.dataset
        for i in range(ds.n):
            yield {"x": ds.xs[i % len(ds.xs)][None], "y": ds.ys[i % len(ds.ys)][None], "p_img": [f"/img_{i}.png"]}


def build_model(network, C):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name=network, weight_type="random",
                                   use_dilated_resnet=True, n_layers=50, width_multiplier=1.0)).cuda()


def round_rate(network, C, H, W, strategy, n_images, batch, dataset_name="cs", ignore_index=None, fused=True, top_n_percent=0.0,
               n_pixels=20, model=None, n_unique=None, rounds=1):
    """-> dict(images_per_s, ms_per_image, peak_gib): one warm-up round, then `rounds` timed rounds over n_images images."""
    ign = C if ignore_index is None else ignore_index
    model = model if model is not None else build_model(network, C)
    ds = _DS(n_images, C, H, W, ign, n_unique)
    prev = ppq.FUSED_LOWRES
    ppq.FUSED_LOWRES = bool(fused)
    try:
        with tempfile.TemporaryDirectory() as td:
            a = Namespace(dataset_name=dataset_name, debug=False, dir_root=td, experim_name="qb", ignore_index=ign, mc_n_steps=20,
                          n_classes=C, n_pixels_by_us=n_pixels, network_name=network, weight_type="random", query_strategy=strategy,
                          reverse_order=False, stride_total=8, top_n_percent=top_n_percent, use_mc_dropout=False, vote_type="hard",
                          query_batch_size=batch)
            qs = ppq.QuerySelector(a, _DL(ds), device=torch.device("cuda"))
            with contextlib.redirect_stdout(io.StringIO()):
                qs(nth_query=1, model=model)          # warm-up: plans, workspaces, pinned buffers
                torch.cuda.synchronize()
                torch.cuda.reset_peak_memory_stats()
                t0 = time.perf_counter()
                for r in range(rounds):
                    qs(nth_query=2 + r, model=model)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / rounds
    finally:
        ppq.FUSED_LOWRES = prev
    return {"images_per_s": n_images / dt, "ms_per_image": dt / n_images * 1e3, "peak_gib": torch.cuda.max_memory_allocated() / 2 ** 30}


def tail_times(model, C, H, W, strategy, batch=1, k=20, reps=5):
    """GPU time (torch events, median of `reps`) of: the forward up to the low-resolution logits; that + the fused interpolate /
    score / top-k launch; the reference-order path (full-resolution branch maps, sums, classifier, logits, scorer).
    -> dict(ms_forward_lowres, ms_fused, ms_reference_order, tail_fused_ms, tail_reference_order_ms): tail = what follows the
    low-resolution logits (fused: timed alone) / what the reference order adds to that same forward (difference of two forwards)."""
    from pixelpick_amd import acquisition as acq
    dev = next(model.parameters()).device
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn((batch, 3, H, W), device=dev, generator=g)
    excl = (torch.rand((batch, H, W), device=dev, generator=g) < 0.05).to(torch.uint8)
    align = bool(getattr(model, "LOWRES_ALIGN_CORNERS", True))
    model.eval()

    def fwd_low():
        return model.forward_lowres(x)

    def fused():
        low, size = model.forward_lowres(x)
        return acq.score_topk_lowres(low, size, excl, strategy, k, align_corners=align)

    def ref_order():
        return acq.score_topk(model(x)["pred"], excl, strategy, k)

    out = {}
    with torch.no_grad():
        for name, fn in (("ms_forward_lowres", fwd_low), ("ms_fused", fused), ("ms_reference_order", ref_order)):
            fn()
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            out[name] = sorted(ts)[len(ts) // 2]
            torch.cuda.empty_cache()
    # the fused tail is ONE launch pair (scorer + candidate merge) of 0.02 - 0.3 ms behind a 10 - 20 ms forward: the difference of two
    # forward timings is noise at that size (it came out negative on some boxes), so it is timed on its own, on logits held still
    with torch.no_grad():
        low, size = model.forward_lowres(x)
        acq.score_topk_lowres(low, size, excl, strategy, k, align_corners=align)
        ts = []
        for _ in range(4 * reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            acq.score_topk_lowres(low, size, excl, strategy, k, align_corners=align)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        del low
    out["tail_fused_ms"] = sorted(ts)[len(ts) // 2]
    out["tail_fused_by_difference_ms"] = out["ms_fused"] - out["ms_forward_lowres"]
    out["tail_reference_order_ms"] = out["ms_reference_order"] - out["ms_forward_lowres"]
    return out


# (label, network, C, H, W, strategy, dataset, ignore, images, batch, distinct images)
CONFIGS = [
    ("configs[2] FPNSeg-ResNet50 256x512 entropy", "FPN", 19, 256, 512, "entropy", "cs", 19, 32, 8, 8),
    ("configs[3] DeepLabv3+-ResNet50 VOC 375x500 (reflect-padded to 376x504) margin", "deeplab_r50", 21, 375, 500, "margin_sampling", "voc", 255, 32, 8, 8),
    ("configs[3] FPNSeg-ResNet50 VOC 375x500 margin", "FPN", 21, 375, 500, "margin_sampling", "voc", 255, 32, 8, 8),
    ("configs[4] FPNSeg-ResNet50 1024x2048 least-confidence", "FPN", 19, 1024, 2048, "least_confidence", "cs", 19, 6, 1, 2),
    ("configs[4] DeepLabv3+-ResNet50 1024x2048 least-confidence", "deeplab_r50", 19, 1024, 2048, "least_confidence", "cs", 19, 6, 1, 2),
]


def main():
    if "--configs" in sys.argv:
        for label, net, C, H, W, st, dsn, ign, n, bs, nu in CONFIGS:
            m = build_model(net, C)
            row = []
            for fused in (True, False):
                r = round_rate(net, C, H, W, st, n, bs, dsn, ign, fused=fused, model=m, n_unique=nu)
                row.append(r)
                torch.cuda.empty_cache()
            f, u = row
            tt = tail_times(m, C, (H + 7) // 8 * 8, (W + 7) // 8 * 8, st, batch=bs)
            print(f"    tail only (GPU time, batch {bs}): fused {tt['tail_fused_ms']:.3f} ms, reference order {tt['tail_reference_order_ms']:.3f} ms "
                  f"(x{tt['tail_reference_order_ms'] / max(tt['tail_fused_ms'], 1e-6):.1f}); forward to the low-resolution logits {tt['ms_forward_lowres']:.2f} ms")
            print(f"{label}: fused tail {f['images_per_s']:7.1f} images/s ({f['ms_per_image']:.2f} ms, peak {f['peak_gib']:.2f} GiB) | "
                  f"reference order {u['images_per_s']:7.1f} images/s ({u['ms_per_image']:.2f} ms, peak {u['peak_gib']:.2f} GiB) | x{f['images_per_s'] / u['images_per_s']:.2f}")
            del m
            torch.cuda.empty_cache()
        return
    N, C, H, W = int(os.environ.get("N", 64)), 19, 256, 512
    model = build_model("deeplab", C)
    for mode in ("k=20", "top5%"):
        for bs in (1, 4, 16):
            r = round_rate("deeplab", C, H, W, "entropy", N, bs, model=model, top_n_percent=0.0 if mode == "k=20" else 0.05,
                           n_pixels=20 if mode == "k=20" else 10)
            print(f"{mode:6s} query_batch_size={bs:2d}: {r['images_per_s']:7.1f} images/s ({r['ms_per_image']:.2f} ms/image)")


if __name__ == "__main__":
    main()
