#!/usr/bin/env python3
"""Target for rocprofv3 --pmc passes: the three SegmentHead 304->256 3x3 kernels (forward, backward-data, weight
gradient) at the BASELINE batch, 10 launches each after a warm-up."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
B, H, W, Cin, Cout, k = 4, 64, 128, 304, 256, 3
x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(k, k, Cin, Cout, device="cuda") * 0.02
y = torch.empty(B, H, W, Cout, device="cuda"); dy = torch.randn(B, H, W, Cout, device="cuda")
dx = torch.empty_like(x); dw = torch.empty_like(w)
ws = torch.empty(L.pp_conv2d_bwd_weight_workspace_bytes(B, H, W, Cin, Cout, k, k, 1, 1, 1), dtype=torch.uint8, device="cuda")
for it in range(13):
    L.pp_conv2d_fwd(x.data_ptr(), Cin, B, H, W, Cin, w.data_ptr(), None, k, k, 1, 1, 1, y.data_ptr(), Cout, Cout, None, 0, st)
    L.pp_conv2d_bwd_data(dy.data_ptr(), Cout, B, H, W, Cout, w.data_ptr(), k, k, 1, 1, 1, dx.data_ptr(), Cin, H, W, Cin, 0, None, 0, st)
    L.pp_conv2d_bwd_weight(x.data_ptr(), Cin, B, H, W, Cin, dy.data_ptr(), Cout, Cout, k, k, 1, 1, 1, dw.data_ptr(), None, ws.data_ptr(), ws.numel(), st)
torch.cuda.synchronize()
