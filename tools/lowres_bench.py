#!/usr/bin/env python3
"""SURVEY.md §8f rank 1 measurement: acquisition from the 1/4-resolution classifier output in one launch
(pp_acq_lowres_score_topk) vs the two-launch path it replaces (pp_bilinear_fwd -> pp_acq_score_topk).
    python tools/lowres_bench.py [B ...]
"""
import sys
import os
os.environ.setdefault("PIXELPICK_KNOBS_BUILD", "1")      # the pp_debug_* planner switches live in the test build only

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import acquisition as acq  # noqa: E402
from pixelpick_amd import engine as E  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def x2_ab():
    """FPNSeg's tail: half-resolution logits, x2 align_corners False - tile height 16 (29 KB patch) vs 32 rows (50.5 KB)."""
    from pixelpick_amd import _lib
    L = _lib.lib()
    for name, C, size, st, B in (("cs", 19, (256, 512), "entropy", 64), ("voc", 21, (376, 504), "margin_sampling", 32),
                                 ("cs-full", 19, (1024, 2048), "least_confidence", 1), ("cs-full", 19, (1024, 2048), "least_confidence", 8)):
        torch.manual_seed(0)
        low = torch.randn(B, size[0] // 2, size[1] // 2, C, device=DEV) * 3
        excl = (torch.rand(B, *size, device=DEV) < 0.05).to(torch.uint8)
        row = []
        for ppt in (0, 4, 8):
            L.pp_debug_set_acq_tuning(0, ppt)
            row.append(timeit(lambda: acq.score_topk_lowres(low, size, excl, st, 20, align_corners=False)))
        L.pp_debug_set_acq_tuning(0, 0)
        npix = B * size[0] * size[1]
        print(f"x2 {name:8s} B={B:3d} {st:18s}: auto {row[0]:.4f} ms ({npix / row[0] / 1e6:.1f} Gpix/s), 16-row tile {row[1]:.4f}, 32-row tile {row[2]:.4f}")


def main():
    if "--x2" in sys.argv:
        return x2_ab()
    Bs = [int(v) for v in sys.argv[1:]] or [1, 8, 64, 256]
    cfgs = [("cs", 19, (64, 128), (256, 512), "entropy"), ("cv", 11, (90, 120), (360, 480), "margin_sampling"),
            ("voc", 21, (80, 80), (320, 320), "margin_sampling"), ("cs-full", 19, (256, 512), (1024, 2048), "least_confidence")]
    print(f"{'cfg':8s} {'B':>4s} {'fused ms':>9s} {'2-launch ms':>11s} {'speed-up':>8s} {'Gpix/s fused':>12s} {'low-res GB/s':>12s}")
    for name, C, lo, size, st in cfgs:
        for B in Bs:
            if name == "cs-full" and B > 16:
                continue
            torch.manual_seed(0)
            low = torch.randn(B, *lo, C, device=DEV) * 3
            excl = (torch.rand(B, *size, device=DEV) < 0.05).to(torch.uint8)
            k = 20

            def fused():
                acq.score_topk_lowres(low, size, excl, st, k)

            def two():
                pred = E.bilinear(E.Tape(False), E.Var(low), size, True, 0.0, out_nchw=True).t
                acq.score_topk(pred, excl, st, k)

            tf, t2 = timeit(fused), timeit(two)
            npix = B * size[0] * size[1]
            print(f"{name:8s} {B:4d} {tf:9.4f} {t2:11.4f} {t2 / tf:8.2f} {npix / tf / 1e6:12.1f} {low.numel() * 4 / tf / 1e6:12.1f}")


if __name__ == "__main__":
    main()
