#!/usr/bin/env python3
"""Which vendor kernels the library-sgemm yardstick of bench.py's roofline_mfma_1x1 runs, and their pure kernel durations next to ours:
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o g -- python tools/lib_gemm_names.py
(the kernel names of the Tensile solutions encode macro tile, MFMA instruction, K split ...)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import _lib  # noqa: E402

SHAPES = [(2048, 1280, 256), (2448, 160, 960), (2048, 960, 160), (2048, 960, 320), (8192, 256, 1024), (8192, 1024, 256), (8192, 2048, 512),
          (32768, 64, 256), (8192, 512, 256), (8192, 1024, 512), (8192, 2048, 256), (8192, 512, 2048), (32768, 256, 64)]


def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for m, k, n in SHAPES:
        x = torch.randn((m, k), device=dev)
        w = torch.randn((k, n), device=dev) * 0.05
        y = torch.empty((m, n), device=dev)
        for _ in range(12):
            torch.matmul(x, w, out=y)
        torch.cuda.synchronize()
        wsb = int(L.pp_conv2d_fwd_workspace_bytes(1, 1, m, k, n, 1, 1, 1, 0, 1))
        ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
        for _ in range(12):
            _lib.check(L.pp_conv2d_fwd(x.data_ptr(), k, 1, 1, m, k, w.data_ptr(), None, 1, 1, 1, 0, 1, y.data_ptr(), n, n,
                                       ws.data_ptr() if wsb else None, wsb, st), "fwd")
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
