#!/usr/bin/env python3
"""Depthwise 3x3 weight gradient on the MobileNetV2 shapes of the BASELINE step (B=4, 256x512, output stride 16): us per call
(partial kernel + combine, 20 back-to-back calls) per geometry word of pp_debug_set_dw_variant, next to the bytes a call has
to read (x + dy once) at 5 TB/s.   python tools/dw_wgrad_bench.py [word ...]"""
import os
os.environ.setdefault("PIXELPICK_KNOBS_BUILD", "1")      # the pp_debug_* planner switches live in the test build only
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
# (name, H, W, C, stride, pad, dil) of the depthwise input; B = 4
LAYERS = [("b1 128x256x32", 128, 256, 32, 1, 1, 1), ("b2 130x258x96 s2", 130, 258, 96, 2, 0, 1), ("b3 64x128x144", 64, 128, 144, 1, 1, 1),
          ("b4 66x130x144 s2", 66, 130, 144, 2, 0, 1), ("b5 32x64x192", 32, 64, 192, 1, 1, 1), ("b7 34x66x192 s2", 34, 66, 192, 2, 0, 1),
          ("b8 16x32x384", 16, 32, 384, 1, 1, 1), ("b12 16x32x576", 16, 32, 576, 1, 1, 1), ("b15 16x32x960 d2", 16, 32, 960, 1, 2, 2),
          ("b16 16x32x960", 16, 32, 960, 1, 1, 1)]


def run(word):
    L.pp_debug_set_dw_variant(word)
    st = torch.cuda.current_stream().cuda_stream
    tot = 0.0
    out = []
    for name, H, W, C, s, p, d in LAYERS:
        B = 4
        Ho, Wo = (H + 2 * p - 2 * d - 1) // s + 1, (W + 2 * p - 2 * d - 1) // s + 1
        x = torch.randn(B, H, W, C, device=dev)
        dy = torch.randn(B, Ho, Wo, C, device=dev)
        dw = torch.empty(3, 3, C, device=dev)
        ws = torch.empty(int(L.pp_colreduce_workspace_bytes(B * Ho * Wo, C)), dtype=torch.uint8, device=dev)

        def fn():
            _lib.check(L.pp_dwconv3x3_bwd_weight(x.data_ptr(), C, B, H, W, C, dy.data_ptr(), C, s, p, d, dw.data_ptr(), ws.data_ptr(), ws.numel(), st), "dw")
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        byt = (x.numel() + dy.numel()) * 4
        out.append(f"{name} {us:5.1f} (bound {byt / 5e6:4.1f})")
        tot += us
    print(f"word {word:#8x}: sum {tot:6.1f} us | " + " | ".join(out))


def main():
    words = [int(w, 0) for w in sys.argv[1:]] or [0] + [(cb << 13) | (mp << 16) for cb in (1, 2, 3, 4) for mp in (1, 2, 0)]
    for w in words:
        run(w)
    L.pp_debug_set_dw_variant(0)


if __name__ == "__main__":
    main()
