#!/usr/bin/env python3
"""Kernel-level timing of conv_x3_kernel variants (pp_debug_set_x3 modes) on the SegmentHead shapes: rocprofv3 gives the per-kernel
durations; this script runs forward, backward-data and the weight gradient of the two layers N times per mode."""
import os, sys
os.environ.setdefault("PIXELPICK_KNOBS_BUILD", "1")      # the pp_debug_* planner switches live in the test build only
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import _lib, engine as E
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
L.pp_debug_set_conv_variant(int(os.environ.get("CONVVAR", "0")))
modes = [int(m) for m in os.environ.get("MODES", "0,1,3,5").split(",")]
shapes = [(4, 64, 128, 304, 256), (4, 64, 128, 256, 256)]
variants = [int(v) for v in os.environ.get("VARS", "0").split(",")]      # pp_debug_set_x3_variant forms of conv_x3_kernel<256,128>
for mode, var in [(m, v) for m in modes for v in variants]:
    L.pp_debug_set_x3(mode)
    L.pp_debug_set_x3_variant(var)
    for (B, H, W, Cin, Cout) in shapes:
        x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(3, 3, Cin, Cout, device="cuda") * 0.02
        if os.environ.get("ZERO"):          # all-zero operands: the same instruction stream at a fraction of the switching power (clock check)
            x.zero_(); w.zero_()
        y = torch.empty(B, H, W, Cout, device="cuda"); dy = torch.randn(B, H, W, Cout, device="cuda"); dx = torch.empty_like(x)
        wf = int(L.pp_conv2d_fwd_workspace_bytes(B, H, W, Cin, Cout, 3, 3, 1, 1, 1)); wb = int(L.pp_conv2d_bwd_data_workspace_bytes(B, H, W, Cin, Cout, 3, 3, 1, 1, 1))
        ww = int(L.pp_conv2d_bwd_weight_workspace_bytes(B, H, W, Cin, Cout, 3, 3, 1, 1, 1)); dw = torch.empty_like(w)
        ws = torch.empty(max(wf, wb, ww, 256), dtype=torch.uint8, device="cuda")
        def fwd(): _lib.check(L.pp_conv2d_fwd(x.data_ptr(), Cin, B, H, W, Cin, w.data_ptr(), None, 3, 3, 1, 1, 1, y.data_ptr(), Cout, Cout, ws.data_ptr(), ws.numel(), st), "f")
        def bwd(): _lib.check(L.pp_conv2d_bwd_data(dy.data_ptr(), Cout, B, H, W, Cout, w.data_ptr(), 3, 3, 1, 1, 1, dx.data_ptr(), Cin, H, W, Cin, 0, ws.data_ptr(), ws.numel(), st), "b")
        def wgr(): _lib.check(L.pp_conv2d_bwd_weight(x.data_ptr(), Cin, B, H, W, Cin, dy.data_ptr(), Cout, Cout, 3, 3, 1, 1, 1, dw.data_ptr(), None, ws.data_ptr(), ws.numel(), st), "w")
        for name, fn in ((("fwd", fwd), ("bwdD", bwd), ("wgrad", wgr)) if not os.environ.get("FWD_ONLY") else (("fwd", fwd),)):
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 20 * 1e3
            gf = 2.0 * B * H * W * Cin * Cout * 9 / 1e9
            if name == "fwd" and os.environ.get("CHECK"):          # same MFMA order in every correct variant: bit-identical to the product kernel
                L.pp_debug_set_x3_variant(0); y.zero_(); fwd(); y0 = y.clone()
                L.pp_debug_set_x3_variant(var); y.zero_(); fwd()
                print(f"   var {var} {Cin}->{Cout}: output {'bit-identical to' if torch.equal(y, y0) else 'DIFFERS from'} variant 0")
            print(f"mode {mode} var {var} {Cin}->{Cout} {name}: {t:7.1f} us per call (incl. operand splits) = {gf / t * 1e3:6.1f} TF-equivalent")
L.pp_debug_set_x3(1)
L.pp_debug_set_x3_variant(0)
