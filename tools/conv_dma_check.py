#!/usr/bin/env python3
"""A/B of the LDS-DMA 128x128 convolution kernel (pp_debug_set_conv_variant: bits 8 / 18 = 128x128 / 64x64 variant off, bit 15 = 128x128 backward-data too) against the register-staged one:
outputs must be bit-identical (same tiles, same MFMA order); prints per-shape timings."""
import os, sys
os.environ.setdefault("PIXELPICK_KNOBS_BUILD", "1")      # the pp_debug_* planner switches live in the test build only
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import _lib, engine as E

DEV = "cuda:0"
L = _lib.lib()


def run(x, w, b, stride, pad, dil, dy):
    tape = E.Tape(True)
    xv = E.Var(x)
    w.grad = None
    w.requires_grad_(True)
    y = E.conv2d(tape, xv, w, b, stride, pad, dil)
    tape.backward(y, dy)
    torch.cuda.synchronize()
    return y.t.clone(), xv.grad.clone(), tape.param_grads[id(w)].clone()


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


shapes = [  # B, H, W, Cin, Cout, k, stride, pad, dil
    (4, 16, 32, 160, 960, 1, 1, 0, 1), (4, 16, 32, 960, 160, 1, 1, 0, 1), (4, 16, 32, 1280, 256, 1, 1, 0, 1), (4, 32, 64, 256, 1024, 1, 1, 0, 1),
    (4, 32, 64, 64, 64, 3, 1, 1, 1), (4, 32, 64, 144, 32, 1, 1, 0, 1),
    (4, 64, 128, 304, 256, 3, 1, 1, 1), (4, 64, 128, 256, 256, 3, 1, 1, 1), (4, 32, 64, 1024, 256, 1, 1, 0, 1),
    (4, 32, 64, 512, 512, 3, 1, 2, 2), (4, 128, 256, 128, 128, 3, 1, 1, 1), (3, 37, 53, 132, 260, 3, 1, 1, 1),
    (4, 64, 128, 256, 512, 1, 2, 0, 1), (2, 45, 61, 136, 192, 3, 2, 1, 1), (4, 16, 32, 320, 256, 3, 1, 12, 12),
]
ok = True
for (B, H, W, Ci, Co, k, st, pad, dil) in shapes:
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Ci, device=DEV)
    w = torch.randn(k, k, Ci, Co, device=DEV) * 0.05
    Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // st + 1, (W + 2 * pad - dil * (k - 1) - 1) // st + 1
    dy = torch.randn(B, Ho, Wo, Co, device=DEV)
    L.pp_debug_set_conv_variant(256 | 262144 | (1 << 20))
    y0, dx0, dw0 = run(x, w, None, st, pad, dil, dy)
    L.pp_debug_set_conv_variant(2 << 20)
    y1, dx1, dw1 = run(x, w, None, st, pad, dil, dy)
    same = torch.equal(y0, y1) and torch.equal(dx0, dx1)
    same_w = torch.equal(dw0, dw1)
    ok &= same_w
    close = torch.allclose(y0, y1, rtol=1e-4, atol=1e-4) and torch.allclose(dx0, dx1, rtol=1e-4, atol=1e-4)   # the 64-deep-K baseline sums in another order
    ok &= close
    fl = 2.0 * B * Ho * Wo * Ci * Co * k * k
    t = {}
    for v in (256 | 262144, 0):
        L.pp_debug_set_conv_variant(v)
        ws, wsn = E._conv_ws(False, x.device, B, H, W, Ci, Co, k, k, st, pad, dil)
        y = torch.empty(B, Ho, Wo, Co, device=DEV)

        def f():
            L.pp_conv2d_fwd(x.data_ptr(), Ci, B, H, W, Ci, w.data_ptr(), None, k, k, st, pad, dil, y.data_ptr(), Co, Co, ws, wsn,
                            _lib.current_stream_ptr())
        t[v] = timeit(f)
    tw = {}
    for v in (1 << 20, 2 << 20):
        L.pp_debug_set_conv_variant(v)
        dwb = torch.empty_like(w)
        nb = L.pp_conv2d_bwd_weight_workspace_bytes(B, H, W, Ci, Co, k, k, st, pad, dil)
        wsb = torch.empty(max(nb, 256), dtype=torch.uint8, device=DEV)

        def g():
            L.pp_conv2d_bwd_weight(x.data_ptr(), Ci, B, H, W, Ci, dy.data_ptr(), Co, Co, k, k, st, pad, dil, dwb.data_ptr(), None,
                                   wsb.data_ptr(), wsb.numel(), _lib.current_stream_ptr())
        tw[v] = timeit(g)
    print(f"{B}x{H}x{W} {Ci}->{Co} k{k} s{st} d{dil}: bit-identical={same} wgrad-identical={same_w} wgrad {tw[1 << 20]:.1f} -> {tw[2 << 20]:.1f} us ({fl / tw[1 << 20] / 1e6:.1f} -> {fl / tw[2 << 20] / 1e6:.1f} TF)  max|dy|={float((y0 - y1).abs().max()):.2e} "
          f"max|ddx|={float((dx0 - dx1).abs().max()):.2e}  fwd {t[256 | 262144]:.1f} -> {t[0]:.1f} us  ({fl / t[256 | 262144] / 1e6:.1f} -> {fl / t[0] / 1e6:.1f} TF)")
L.pp_debug_set_conv_variant(0)
print("ALL EQUAL (bit-identical unless the baseline used the 64-deep K step)" if ok else "MISMATCH")
