#!/usr/bin/env python3
"""Per-layer table of the dense convolutions in one DeepLabv3+-MNv2 (or FPN) train step at the BASELINE
config: shape, call count, and µs / TFLOP/s of forward, backward-data and backward-weight, each timed alone
through the C ABI.  Tells which layer shapes the implicit-GEMM kernels serve badly."""
import os, sys, warnings, collections
os.environ.setdefault("PIXELPICK_KNOBS_BUILD", "1")      # the pp_debug_* planner switches live in the test build only
from argparse import Namespace
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import engine, _lib
from pixelpick_amd.utils.utils import get_model
from pixelpick_amd.trainer import FlatTrainer


def main():
    net = os.environ.get("NET", "deeplab")
    B, H, W, C = int(os.environ.get("B", 4)), 256, 512, 19
    args = Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name=net, weight_type="random",
                     n_layers=50, use_softmax=True, use_dilated_resnet=True, width_multiplier=1.0)
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = get_model(args).cuda().train()
    tr = FlatTrainer(m, ignore_index=C)
    x = torch.randn(B, 3, H, W, device="cuda")
    y = torch.full((B, H, W), C, dtype=torch.int64, device="cuda")
    y[:, ::37, ::41] = 3
    seen = collections.OrderedDict()
    orig = engine.conv2d

    def rec(tape, xv, w, bias, stride=1, pad=0, dil=1, dst=None, **kw):
        Bn, Hh, Ww, Cin, _ = engine._geom(xv.t)
        key = (Bn, Hh, Ww, Cin, w.shape[3], w.shape[0], w.shape[1], stride, pad, dil, bool(xv.needs_grad))
        seen[key] = seen.get(key, 0) + 1
        return orig(tape, xv, w, bias, stride, pad, dil, dst, **kw)

    engine.conv2d = rec
    for mod in list(sys.modules.values()):
        if mod and getattr(mod, "__name__", "").startswith("pixelpick_amd") and getattr(mod, "conv2d", None) is orig:
            mod.conv2d = rec
    tr.train_step(x, y)
    torch.cuda.synchronize()
    engine.conv2d = orig
    L = _lib.lib()
    L.pp_debug_set_conv_variant(int(os.environ.get("CONV_VARIANT", 0)))      # A/B of the kernel choices (conv_igemm.hip)
    if os.environ.get("WGRAD_TARGET"):
        L.pp_debug_set_wgrad_target(int(os.environ["WGRAD_TARGET"]))
    if os.environ.get("SPLITK"):           # pp_debug_set_splitk word: tiles_threshold | target << 10 | min_k_steps << 20 | min_steps_per_slice << 26
        L.pp_debug_set_splitk(int(os.environ["SPLITK"]))
    only = os.environ.get("ONLY")          # e.g. ONLY=64x128x3: rows of that H x W x kernel only
    st = torch.cuda.current_stream().cuda_stream

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

    tot = [0.0, 0.0, 0.0]
    print(f"{'B,H,W':>12s} {'Cin':>5s} {'Cout':>5s} {'k':>3s} {'s':>2s} {'d':>3s} {'n':>3s} {'GF':>7s} | {'fwd us':>8s} {'TF':>6s} | {'bwdD us':>8s} {'TF':>6s} | {'bwdW us':>8s} {'TF':>6s}")
    for (Bn, Hh, Ww, Cin, Cout, kh, kw, s, p, d, ng), cnt in seen.items():
        if only and only != f"{Hh}x{Ww}x{kh}":
            continue
        Ho, Wo = engine.out_size(Hh, kh, s, p, d), engine.out_size(Ww, kw, s, p, d)
        xt = torch.randn(Bn, Hh, Ww, Cin, device="cuda")
        wt = torch.randn(kh, kw, Cin, Cout, device="cuda")
        yt = torch.empty(Bn, Ho, Wo, Cout, device="cuda")
        dy = torch.randn(Bn, Ho, Wo, Cout, device="cuda")
        dx = torch.empty_like(xt)
        dw = torch.empty_like(wt)
        ws = torch.empty(max(1, L.pp_conv2d_bwd_weight_workspace_bytes(Bn, Hh, Ww, Cin, Cout, kh, kw, s, p, d)), dtype=torch.uint8, device="cuda")
        wsf = torch.empty(max(1, L.pp_conv2d_fwd_workspace_bytes(Bn, Hh, Ww, Cin, Cout, kh, kw, s, p, d)), dtype=torch.uint8, device="cuda")
        wsd = torch.empty(max(1, L.pp_conv2d_bwd_data_workspace_bytes(Bn, Hh, Ww, Cin, Cout, kh, kw, s, p, d)), dtype=torch.uint8, device="cuda")
        gf = 2.0 * Bn * Ho * Wo * Cin * Cout * kh * kw / 1e9
        tf = timeit(lambda: L.pp_conv2d_fwd(xt.data_ptr(), Cin, Bn, Hh, Ww, Cin, wt.data_ptr(), None, kh, kw, s, p, d, yt.data_ptr(), Cout, Cout, wsf.data_ptr(), wsf.numel(), st))
        td = timeit(lambda: L.pp_conv2d_bwd_data(dy.data_ptr(), Cout, Bn, Ho, Wo, Cout, wt.data_ptr(), kh, kw, s, p, d, dx.data_ptr(), Cin, Hh, Ww, Cin, 0, wsd.data_ptr(), wsd.numel(), st)) if ng else 0.0
        tw = timeit(lambda: L.pp_conv2d_bwd_weight(xt.data_ptr(), Cin, Bn, Hh, Ww, Cin, dy.data_ptr(), Cout, Cout, kh, kw, s, p, d, dw.data_ptr(), None, ws.data_ptr(), ws.numel(), st))
        tot[0] += tf * cnt; tot[1] += td * cnt; tot[2] += tw * cnt
        f = lambda t: (gf / t * 1e3) if t else 0.0
        print(f"{Bn:2d},{Hh:4d},{Ww:4d} {Cin:5d} {Cout:5d} {kh:3d} {s:2d} {d:3d} {cnt:3d} {gf:7.2f} | {tf:8.1f} {f(tf):6.1f} | {td:8.1f} {f(td):6.1f} | {tw:8.1f} {f(tw):6.1f}")
    print(f"per step: fwd {tot[0]/1e3:.2f} ms, bwd-data {tot[1]/1e3:.2f} ms, bwd-weight {tot[2]/1e3:.2f} ms")


if __name__ == "__main__":
    main()
