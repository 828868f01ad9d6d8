#!/usr/bin/env python3
"""The pointwise GEMM kernel (csrc/gemm_pw.hip) on the graded 1x1 shapes and the other ResNet50 pointwise layers at the BASELINE batch: us per
call (HIP-event pairs, median of 20: ~5 us of launch in every figure) for each forced tile form, the planner's rule, the kernels it replaces
(pp_debug_set_gemm_pw(0)) and the vendor sgemm.  GPU box:  python tools/gemm_pw_bench.py"""
import os
os.environ.setdefault("PIXELPICK_KNOBS_BUILD", "1")      # the pp_debug_* planner switches live in the test build only
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import _lib  # noqa: E402

SHAPES = [("R50 256->1024 @8192", 8192, 256, 1024), ("R50 1024->256 @8192", 8192, 1024, 256), ("R50 2048->512 @8192", 8192, 2048, 512),
          ("R50 64->256 @32768", 32768, 64, 256), ("R50 256->64 @32768", 32768, 256, 64), ("R50 64->64 @32768", 32768, 64, 64),
          ("R50 512->256 @8192", 8192, 512, 256), ("R50 1024->512 @8192", 8192, 1024, 512), ("R50 512->2048 @8192", 8192, 512, 2048),
          ("R50 1024->2048 @8192", 8192, 1024, 2048), ("R50 256->128 @8192", 8192, 256, 128), ("R50 512->128 @8192", 8192, 512, 128),
          ("R50 128->512 @8192", 8192, 128, 512), ("ASPP-R50 2048->256 @8192", 8192, 2048, 256), ("FPN 2048->256 @8192", 8192, 2048, 256),
          ("FPN 256->256 @32768", 32768, 256, 256), ("x16 256->1024 @131072", 131072, 256, 1024),
          # the 1/16-resolution layers of the MobileNetV2 model (2048 rows at B = 4: below the planner's row limit - forced forms show why)
          ("MNv2 960->320 @2048", 2048, 960, 320), ("ASPP fuse 1280->256 @2048", 2048, 1280, 256), ("MNv2 160->960 @2448", 2448, 160, 960),
          ("MNv2 960->160 @2048", 2048, 960, 160), ("ASPP 320->256 @2048", 2048, 320, 256)]
if os.environ.get("ONLY"):
    SHAPES = [s_ for s_ in SHAPES if os.environ["ONLY"] in s_[0]]
FORMS = ["128x256", "256x128", "128x128", "64x128", "128x64", "64x64"]


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
    print("shape                     | GFLOP | " + " ".join(f"{f:>7s}" for f in FORMS) + " |    rule | replaced |  vendor | rule TF (of 157.3) | rule / vendor")
    for name, m, k, n in SHAPES:
        x = torch.randn((m, k), device=dev)
        w = torch.randn((k, n), device=dev) * 0.05
        y = torch.empty((m, n), device=dev)

        def run():
            wsb = int(L.pp_conv2d_fwd_workspace_bytes(1, 1, m, k, n, 1, 1, 1, 0, 1))
            ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)

            def f():
                _lib.check(L.pp_conv2d_fwd(x.data_ptr(), k, 1, 1, m, k, w.data_ptr(), None, 1, 1, 1, 0, 1, y.data_ptr(), n, n,
                                           ws.data_ptr() if wsb else None, wsb, st), "fwd")
            return timeit(f)
        ts = []
        for form in range(6):
            L.pp_debug_set_gemm_pw((2 + form) | (1 << 4))
            ts.append(run())
        L.pp_debug_set_gemm_pw(1)
        L.pp_debug_set_x3(0)                      # (the fp32 kernels on both sides: the bf16x3 rule has its own record)
        rule = run()
        L.pp_debug_set_gemm_pw(0)
        old = run()
        L.pp_debug_set_x3(1)
        L.pp_debug_set_gemm_pw(1)
        lib = timeit(lambda: torch.matmul(x, w, out=y))
        # backward-data (dX = dY x W^T): rule vs replaced
        dx = torch.empty((m, k), device=dev)

        def runb():
            wsb = int(L.pp_conv2d_bwd_data_workspace_bytes(1, 1, m, k, n, 1, 1, 1, 0, 1))
            ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)

            def f():
                _lib.check(L.pp_conv2d_bwd_data(y.data_ptr(), n, 1, 1, m, n, w.data_ptr(), 1, 1, 1, 0, 1, dx.data_ptr(), k, 1, m, k, 0,
                                                ws.data_ptr() if wsb else None, wsb, st), "bwd")
            return timeit(f)
        L.pp_debug_set_x3(0)
        brule = runb()
        L.pp_debug_set_gemm_pw(0)
        bold = runb()
        L.pp_debug_set_gemm_pw(1)
        L.pp_debug_set_x3(1)
        gf = 2.0 * m * k * n / 1e9
        print(f"{name:25s} | {gf:5.2f} | " + " ".join(f"{t:7.1f}" for t in ts) + f" | {rule:7.1f} | {old:8.1f} | {lib:7.1f} | {gf / rule * 1e3:6.1f} ({gf / rule * 1e3 / 157.3:4.2f}) | {lib / rule:5.2f} | bwd-data {brule:7.1f} (replaced {bold:7.1f})")


if __name__ == "__main__":
    main()
