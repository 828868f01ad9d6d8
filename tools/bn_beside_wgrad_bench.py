#!/usr/bin/env python3
"""The SegmentHead BatchNorm backward (33.5 MB map) alone and BESIDE the SegmentHead weight gradient on a second stream - the
situation of the train step's backward (profiles/r02_train_ablation.txt).  Prints the BatchNorm launch's duration (HIP events on its
stream) for several grid targets, and the weight gradient's own duration with / without the BatchNorm next to it."""
import os
os.environ.setdefault("PIXELPICK_KNOBS_BUILD", "1")      # the pp_debug_* planner switches live in the test build only
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import _lib  # noqa: E402
from pixelpick_amd import engine as E  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
B, H, W, Cin, Cout = 4, 64, 128, int(os.environ.get("CIN", 256)), 256
M = B * H * W
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
x, dy = torch.randn(M, Cout, device=dev), torch.randn(M, Cout, device=dev)
gamma, beta = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
mean, invstd = torch.randn(Cout, device=dev) * 0.1, torch.rand(Cout, device=dev) + 0.5
dg, db = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
dx = torch.empty_like(x)
sync, ws = E._bn_exchange(dev)
xin = torch.randn(B, H, W, Cin, device=dev)
dyc = torch.randn(B, H, W, Cout, device=dev)
dw = torch.empty(3, 3, Cin, Cout, device=dev)
wsw = torch.empty(L.pp_conv2d_bwd_weight_workspace_bytes(B, H, W, Cin, Cout, 3, 3, 1, 1, 1), dtype=torch.uint8, device=dev)


def bn():
    _lib.check(L.pp_bn_bwd_fused(x.data_ptr(), Cout, dy.data_ptr(), Cout, None, Cout, 1, M, Cout, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                 dg.data_ptr(), db.data_ptr(), dx.data_ptr(), Cout, None, 0, 1.0, beta.data_ptr(), ws.data_ptr(), ws.numel(),
                                 sync.data_ptr(), sync.numel(), main.cuda_stream), "bn")


def wgrad():
    _lib.check(L.pp_conv2d_bwd_weight(xin.data_ptr(), Cin, B, H, W, Cin, dyc.data_ptr(), Cout, Cout, 3, 3, 1, 1, 1, dw.data_ptr(), None,
                                      wsw.data_ptr(), wsw.numel(), side.cuda_stream), "wgrad")


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def run(beside, delay_us):
    tb, tw = [], []
    for _ in range(9):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if beside:
            c.record(side)
            wgrad()
            d.record(side)
            torch.cuda._sleep(int(delay_us * 2100))          # ~cycles: let the weight gradient get going first
        a.record(main)
        bn()
        b.record(main)
        torch.cuda.synchronize()
        tb.append(a.elapsed_time(b) * 1e3)
        if beside:
            tw.append(c.elapsed_time(d) * 1e3)
    return med(tb), (med(tw) if tw else 0.0)


for target in [int(v) for v in (sys.argv[1:] or ["256", "384", "128"])]:
    L.pp_debug_set_bn_target(target)
    alone, _ = run(False, 0)
    b100, w100 = run(True, 100)
    b300, w300 = run(True, 300)
    print(f"bn grid target {target:4d}: alone {alone:6.1f} us | beside the weight gradient, launched 100 us into it: {b100:6.1f} us (wgrad {w100:6.1f}) | "
          f"300 us into it: {b300:6.1f} us (wgrad {w300:6.1f})")
torch.cuda.synchronize()
wa = []
for _ in range(5):
    c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c.record(side); wgrad(); d.record(side); torch.cuda.synchronize(); wa.append(c.elapsed_time(d) * 1e3)
print(f"weight gradient alone {med(wa):6.1f} us")
