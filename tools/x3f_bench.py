#!/usr/bin/env python3
"""A/B of the in-kernel activation split (csrc/conv_x3f.hip) on the ResNet50 / head layers at the BASELINE batch: forward and backward-data
per layer with the in-kernel split (pp_debug_set_x3f(1)), without it (the product's plan - fp32-MFMA kernels or x3_split +
conv_x3_kernel) and, where conv_x3_kernel applies, with the caller holding the planes (kernel only); vendor sgemm beside the 1x1 layers.
GPU box:  python tools/x3f_bench.py"""
import os
os.environ.setdefault("PIXELPICK_KNOBS_BUILD", "1")      # the pp_debug_* planner switches live in the test build only
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import _lib  # noqa: E402

# name, B, H, W, Cin, Cout, k, stride, pad, dil
LAYERS = [("R50 l1 64->64 3x3 @64x128", 4, 64, 128, 64, 64, 3, 1, 1, 1), ("R50 l1 256->64 @64x128", 4, 64, 128, 256, 64, 1, 1, 0, 1),
          ("R50 l2 256->128 @32x64", 4, 32, 64, 256, 128, 1, 1, 0, 1), ("R50 l2 128->128 3x3 @32x64", 4, 32, 64, 128, 128, 3, 1, 1, 1),
          ("R50 l2 512->128 @32x64", 4, 32, 64, 512, 128, 1, 1, 0, 1), ("R50 l3 512->256 @32x64", 4, 32, 64, 512, 256, 1, 1, 0, 1),
          ("R50 l3 256->256 3x3 d2 @32x64", 4, 32, 64, 256, 256, 3, 1, 2, 2), ("R50 l3 256->1024 @32x64", 4, 32, 64, 256, 1024, 1, 1, 0, 1),
          ("R50 l3 1024->256 @32x64", 4, 32, 64, 1024, 256, 1, 1, 0, 1), ("R50 l4 1024->512 @32x64", 4, 32, 64, 1024, 512, 1, 1, 0, 1),
          ("R50 l4 512->512 3x3 d4 @32x64", 4, 32, 64, 512, 512, 3, 1, 4, 4), ("R50 l4 512->2048 @32x64", 4, 32, 64, 512, 2048, 1, 1, 0, 1),
          ("R50 l4 2048->512 @32x64", 4, 32, 64, 2048, 512, 1, 1, 0, 1), ("R50 l4 1024->2048 @32x64", 4, 32, 64, 1024, 2048, 1, 1, 0, 1),
          ("FPN 256->256 3x3 @32x64", 4, 32, 64, 256, 256, 3, 1, 1, 1), ("FPN 256->128 3x3 @64x128", 4, 64, 128, 256, 128, 3, 1, 1, 1),
          ("FPN 128->128 3x3 @128x256", 4, 128, 256, 128, 128, 3, 1, 1, 1), ("ASPP-R50 2048->256 @32x64", 4, 32, 64, 2048, 256, 1, 1, 0, 1),
          ("SegmentHead 304->256 3x3 @64x128", 4, 64, 128, 304, 256, 3, 1, 1, 1), ("SegmentHead 256->256 3x3 @64x128", 4, 64, 128, 256, 256, 3, 1, 1, 1)]


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


def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    only = sys.argv[1:]
    print("layer | GFLOP | fwd us: in-kernel split, round-5 plan, planes held (kernel only) | bwd-data us: same three | vendor sgemm us (1x1) | fwd TF in-kernel")
    for name, B, H, W, ci, co, k, s, p, d in LAYERS:
        if only and not any(o in name for o in only):
            continue
        Ho, Wo = (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1
        x = torch.randn((B, H, W, ci), device=dev)
        w = torch.randn((k, k, ci, co), device=dev) * 0.05
        y = torch.empty((B, Ho, Wo, co), device=dev)
        dy = torch.randn((B, Ho, Wo, co), device=dev)
        dx = torch.empty((B, H, W, ci), device=dev)
        gf = 2.0 * B * Ho * Wo * ci * co * k * k / 1e9
        res = []
        for word in (1, 0):
            L.pp_debug_set_x3f(word)
            wsb = int(L.pp_conv2d_fwd_workspace_bytes(B, H, W, ci, co, k, k, s, p, d))
            ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
            wsd = int(L.pp_conv2d_bwd_data_workspace_bytes(B, H, W, ci, co, k, k, s, p, d))
            wd = torch.empty(max(wsd, 256), dtype=torch.uint8, device=dev)

            def fwd():
                _lib.check(L.pp_conv2d_fwd(x.data_ptr(), ci, B, H, W, ci, w.data_ptr(), None, k, k, s, p, d, y.data_ptr(), co, co,
                                           ws.data_ptr() if wsb else None, wsb, st), "fwd")

            def bwd():
                _lib.check(L.pp_conv2d_bwd_data(dy.data_ptr(), co, B, Ho, Wo, co, w.data_ptr(), k, k, s, p, d, dx.data_ptr(), ci, H, W, ci, 0,
                                                wd.data_ptr() if wsd else None, wsd, st), "bwd")
            res.append((timeit(fwd), timeit(bwd)))
        L.pp_debug_set_x3f(0)
        held = ("   -  ", "   -  ")
        nbx = int(L.pp_conv2d_x3_planes_bytes(0, B, H, W, ci, co, k, k, s, p, d))
        if nbx:
            xp = torch.empty(nbx, dtype=torch.uint8, device=dev)
            wp = torch.empty(int(L.pp_x3_weight_planes_bytes(k * k, ci, co, 1)), dtype=torch.uint8, device=dev)
            _lib.check(L.pp_x3_split(x.data_ptr(), ci, B * H * W, ci, xp.data_ptr(), nbx, st), "split")
            _lib.check(L.pp_x3_split_weights(w.data_ptr(), k * k, ci, co, 1, wp.data_ptr(), wp.numel(), st), "splitw")
            wsb = int(L.pp_conv2d_fwd_workspace_bytes(B, H, W, ci, co, k, k, s, p, d))
            ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)

            def fwdp():
                _lib.check(L.pp_conv2d_fwd_pre2(x.data_ptr(), ci, B, H, W, ci, w.data_ptr(), None, k, k, s, p, d, y.data_ptr(), co, co,
                                                ws.data_ptr(), wsb, xp.data_ptr(), wp.data_ptr(), st), "fwd_pre2")
            held = (f"{timeit(fwdp):6.1f}", "   -  ")
        lib = ""
        if k == 1 and s == 1:
            xm, wm, ym = x.view(-1, ci), w.view(ci, co), torch.empty((B * H * W, co), device=dev)
            lib = f"{timeit(lambda: torch.matmul(xm, wm, out=ym)):6.1f}"
        print(f"{name:34s} | {gf:6.2f} | {res[0][0]:6.1f} {res[1][0]:6.1f} {held[0]} | {res[0][1]:6.1f} {res[1][1]:6.1f} | {lib:>6s} | {gf / res[0][0] * 1e-3 * 1e3:6.1f}")


if __name__ == "__main__":
    main()
