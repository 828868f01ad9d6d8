#!/usr/bin/env python3
"""Per-layer pieces of the BatchNorm-apply-on-load experiment (engine._BN_ON_LOAD), each timed alone (20 back-to-back calls):
   today:     [pw conv] -> single-launch BatchNorm+ReLU6 -> depthwise 3x3
   on load:   [pw conv + statistics epilogue] -> finalize -> depthwise 3x3 with input affine (+ statistics for the next BatchNorm)
for the expand -> BN -> dw segment of MobileNetV2 blocks at three map sizes (mobilenet_v2.py:42-52)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import _lib, engine as E

L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
dev = "cuda:0"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'map (B,H,W,Cin->C)':28s} | conv  conv+stats | BN fused | finalize (rows) | dw x4/plain  dw fused+affine  +stats (rows)")
for (B, H, W, Cin, C, stride) in [(4, 18, 34, 160, 960, 1), (4, 18, 34, 64, 384, 1), (4, 34, 66, 32, 192, 1), (4, 66, 130, 24, 144, 1),
                                  (4, 130, 258, 16, 96, 2)]:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(1, 1, Cin, C, device=dev) * 0.05
    y = torch.empty(B, H, W, C, device=dev)
    M = B * H * W
    wsn = int(L.pp_conv2d_fwd_workspace_bytes(B, H, W, Cin, C, 1, 1, 1, 0, 1))
    ws = torch.empty(max(wsn, 256), dtype=torch.uint8, device=dev)
    rows = int(L.pp_conv2d_fwd_stats_rows(B, H, W, Cin, C, 1, 1, 1, 0, 1))
    stats = torch.empty(max(rows, 1) * 2 * C, device=dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, invstd, scale, shift = (torch.empty(C, device=dev) for _ in range(4))
    t_conv = timeit(lambda: L.pp_conv2d_fwd(x.data_ptr(), Cin, B, H, W, Cin, w.data_ptr(), None, 1, 1, 1, 0, 1, y.data_ptr(), C, C,
                                            ws.data_ptr() if wsn else None, wsn, st))
    t_cs = timeit(lambda: L.pp_conv2d_fwd_stats(x.data_ptr(), Cin, B, H, W, Cin, w.data_ptr(), None, 1, 1, 1, 0, 1, y.data_ptr(), C, C,
                                                ws.data_ptr() if wsn else None, wsn, stats.data_ptr(), stats.numel(), st)) if rows > 0 else float("nan")
    sync, xws = E._bn_exchange(torch.device(dev))
    yb = torch.empty_like(y)
    t_bn = timeit(lambda: L.pp_bn_train_fwd_fused(y.data_ptr(), C, M, C, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(),
                                                  mean.data_ptr(), invstd.data_ptr(), None, 0, 2, 0.0, 0, None, yb.data_ptr(), C, xws.data_ptr(),
                                                  xws.numel(), sync.data_ptr(), sync.numel(), st))
    t_fin = timeit(lambda: L.pp_bn_finalize_partials(stats.data_ptr(), rows, M, C, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, rm.data_ptr(),
                                                     rv.data_ptr(), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), st))
    wd = torch.randn(3, 3, C, device=dev)
    Ho, Wo = E.out_size(H, 3, stride, 0, 1), E.out_size(W, 3, stride, 0, 1)
    yd = torch.empty(B, Ho, Wo, C, device=dev)
    drows = int(L.pp_dwconv3x3_fwd_stats_rows(B, H, W, C, stride, 0, 1))
    dstats = torch.empty(drows * 2 * C, device=dev)
    t_dw = timeit(lambda: L.pp_dwconv3x3_fwd(yb.data_ptr(), C, B, H, W, C, wd.data_ptr(), stride, 0, 1, yd.data_ptr(), C, st))
    t_dwa = timeit(lambda: L.pp_dwconv3x3_fwd_fused(y.data_ptr(), C, B, H, W, C, wd.data_ptr(), stride, 0, 1, scale.data_ptr(), shift.data_ptr(), 2,
                                                    yd.data_ptr(), C, None, 0, st))
    t_dws = timeit(lambda: L.pp_dwconv3x3_fwd_fused(y.data_ptr(), C, B, H, W, C, wd.data_ptr(), stride, 0, 1, scale.data_ptr(), shift.data_ptr(), 2,
                                                    yd.data_ptr(), C, dstats.data_ptr(), dstats.numel(), st))
    print(f"{str((B, H, W, Cin, C)):28s} | {t_conv:5.1f} {t_cs:6.1f}      | {t_bn:6.1f}   | {t_fin:5.1f} ({rows:5d})   | {t_dw:6.1f}       {t_dwa:6.1f}        "
          f"{t_dws:6.1f} ({drows})   today {t_conv + t_bn + t_dw:6.1f}  on-load {t_cs + t_fin + t_dws:6.1f}")
