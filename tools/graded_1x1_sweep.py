#!/usr/bin/env python3
"""The graded pointwise shapes (bench.py roofline_mfma_1x1) under the library's planning knobs: which existing kernel / tile is fastest
for each, against the vendor sgemm.  GPU box:  python tools/graded_1x1_sweep.py"""
import os, sys
os.environ.setdefault("PIXELPICK_KNOBS_BUILD", "1")      # the pp_debug_* planner switches live in the test build only
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import _lib

SHAPES = [("MNv2 960->320 @2048", 4, 16, 32, 960, 320), ("R50 256->1024 @8192", 4, 32, 64, 256, 1024), ("R50 1024->256 @8192", 4, 32, 64, 1024, 256),
          ("R50 2048->512 @8192", 4, 32, 64, 2048, 512), ("R50 64->256 @32768", 4, 64, 128, 64, 256), ("R50 256->64 @32768", 4, 64, 128, 256, 64),
          ("R50 512->2048 @8192", 4, 32, 64, 512, 2048), ("R50 1024->2048 @8192", 4, 32, 64, 1024, 2048)]
# name -> (thresholds word, variant word, x3 word)
KNOBS = {"default": (0, 0, 1), "128-tiles always": (1, 0, 1), "64x64 always": (4095, 0, 1), "128x64 tiles": (1, 2, 1), "128 bk32": (1, 4096, 1),
         "128 no-dma": (1, 256, 1), "64x64 no-dma": (4095, 262144, 1), "no split-K": (0, 64, 1), "no ksplit": (0, 1 << 25, 1),
         "x3 from 1 GF / 32 tiles": (0, 0, 1 | (7 << 9) | (3 << 12)), "x3 from 4 GF / 128 tiles": (0, 0, 1 | (4 << 9) | (1 << 12)), "x3 off": (0, 0, 0)}


def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for name, B, H, W, ci, co in SHAPES:
        x = torch.randn((B, H, W, ci), device=dev)
        w = torch.randn((1, 1, ci, co), device=dev) * 0.05
        y = torch.empty((B, H, W, co), device=dev)
        m = B * H * W
        xm, wm, ym = x.view(m, ci), w.view(ci, co), torch.empty((m, co), device=dev)
        def lib():
            torch.matmul(xm, wm, out=ym)
        def timeit(fn, n=20):
            for _ in range(5):
                fn()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            torch.cuda.synchronize()
            for a, b in evs:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            t = sorted(a.elapsed_time(b) for a, b in evs)
            return t[len(t) // 2] * 1e3
        t_lib = timeit(lib)
        print(f"{name}: library sgemm {t_lib:.1f} us")
        for kn, (thr, var, x3) in KNOBS.items():
            L.pp_debug_set_conv_thresholds(thr); L.pp_debug_set_conv_variant(var); L.pp_debug_set_x3(x3)
            wsb = int(L.pp_conv2d_fwd_workspace_bytes(B, H, W, ci, co, 1, 1, 1, 0, 1))
            ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
            def ours():
                rc = L.pp_conv2d_fwd(x.data_ptr(), ci, B, H, W, ci, w.data_ptr(), None, 1, 1, 1, 0, 1, y.data_ptr(), co, co,
                                     ws.data_ptr() if wsb else None, wsb, st)
                _lib.check(rc, "pp_conv2d_fwd")
            t = timeit(ours)
            print(f"    {kn:28s} {t:7.1f} us   {t_lib / t:5.2f} x library")
        L.pp_debug_set_conv_thresholds(0); L.pp_debug_set_conv_variant(0); L.pp_debug_set_x3(1)


if __name__ == "__main__":
    main()
