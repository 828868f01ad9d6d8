#!/usr/bin/env python3
"""The FPN decoder's elementwise kernels in isolation (20 back-to-back calls, us per call) against the bytes they move at 5.5 TB/s:
add2d (top-down sums, decoders.py:82,96), bilinear x2 / x4 / x8 upsample forward and backward (decoders.py:36-55,74)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import engine as E  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    tape = E.Tape(False)
    for (B, H, W, C) in ((4, 64, 128, 256), (4, 64, 128, 128), (4, 32, 64, 256), (4, 16, 32, 256)):
        x, y = torch.randn(B, H, W, C, device=dev), torch.randn(B, H, W, C, device=dev)
        us = t(lambda: E.add(tape, E.Var(x), E.Var(y)))
        byt = 3 * x.numel() * 4
        print(f"add2d   [{B},{H},{W},{C}]            {us:6.1f} us   ({byt / 1e6:5.1f} MB: {byt / 5.5e6:5.1f} us at 5.5 TB/s)")
    for (B, h, w, C, s) in ((4, 32, 64, 128, 2), (4, 16, 32, 128, 2), (4, 16, 32, 128, 4), (4, 8, 16, 128, 8), (4, 32, 64, 256, 2), (4, 16, 32, 256, 2)):
        x = torch.randn(B, h, w, C, device=dev)
        H, W = h * s, w * s
        us = t(lambda: E.bilinear(tape, E.Var(x), (H, W), False, 0.0))
        byt = (x.numel() + B * H * W * C) * 4
        tp = E.Tape()
        xv = E.Var(x)
        yv = E.bilinear(tp, xv, (H, W), False, 0.0)
        dy = torch.randn(B, H, W, C, device=dev)
        node = tp.nodes[-1]
        usb = t(lambda: node[0](tp, dy, *node[1]))
        print(f"bilinear [{B},{h},{w},{C}] x{s}  fwd {us:6.1f} us  bwd {usb:6.1f} us   ({byt / 1e6:5.1f} MB: {byt / 5.5e6:5.1f} us at 5.5 TB/s)")


if __name__ == "__main__":
    main()
