"""Per-layer timing of a RECORDED train step: every C-ABI call of FlatTrainer.enable_replay()'s launch plan can be re-issued on its
own with the very arguments (addresses inside the plan's private memory pool) the step uses, so a layer is measured on its real
operands, in isolation, warm (right after a launch that touched the same tensors - the case inside the step) and cold (a 512 MiB
write in between evicts L2 / the 256 MiB MALL).  tools/dw_bench.py and bench.py's `roofline_hbm_depthwise` use it for the
depthwise layers of MobileNetV2 (mobilenet_v2.py:38,52): bandwidth-bound kernels, priced against the HBM roofline."""
import ctypes
from typing import Callable, Dict, List, Optional

import torch


def _ival(a) -> int:
    return int(a.value) if isinstance(a, ctypes._SimpleCData) and a.value is not None else 0


# entry point -> geometry of a depthwise 3x3 call: positions of (B, H, W, C) of the INPUT map and of (stride, pad, dil)
_DW_GEOM = {
    "pp_dwconv3x3_fwd": ((2, 3, 4, 5), (7, 8, 9)),
    "pp_dwconv3x3_fwd_fused": ((2, 3, 4, 5), (7, 8, 9)),
    "pp_dwconv3x3_bn_train_fwd_fused": ((2, 3, 4, 5), (7, 8, 9)),
    "pp_dwconv3x3_bwd_data": ((2, 3, 4, 5), (7, 8, 9)),
    "pp_dwconv3x3_bwd_weight": ((2, 3, 4, 5), (8, 9, 10)),
    "pp_dwconv3x3_bwd_weight_partials": ((2, 3, 4, 5), (8, 9, 10)),
    "pp_dwconv3x3_bwd_weight_affine_in": ((2, 3, 4, 5), (11, 12, 13)),
}


def depthwise_calls(plan) -> List[Dict]:
    """The depthwise launches of a recorded step in launch order: entry point, geometry, algorithmic bytes (each tensor once)."""
    out = []
    for i, (fn, args) in enumerate(plan.calls):
        name = getattr(fn, "__name__", "")
        if name not in _DW_GEOM:
            continue
        (ib, ih, iw, ic), (isd, ipd, idl) = _DW_GEOM[name]
        B, H, W, C = (_ival(args[j]) for j in (ib, ih, iw, ic))
        s, p, d = (_ival(args[j]) for j in (isd, ipd, idl))
        Ho, Wo = (H + 2 * p - 2 * d - 1) // s + 1, (W + 2 * p - 2 * d - 1) // s + 1
        s_in, s_out = 4 * B * H * W * C, 4 * B * Ho * Wo * C
        if name == "pp_dwconv3x3_bwd_data":
            kind, read, write = "bwd_data", s_out, s_in                 # reads dy [Ho, Wo], writes dx [H, W]
        elif "bwd_weight" in name:
            kind, read, write = "bwd_weight", s_in + s_out, 0           # reads x and dy once, writes 9 C floats (+ partials)
        else:
            kind, read, write = "fwd", s_in, s_out
            if name == "pp_dwconv3x3_bn_train_fwd_fused":
                write += s_out                                          # raw convolution (the BatchNorm backward's input) + the activated output
        out.append({"index": i, "entry": name, "kind": kind, "B": B, "H": H, "W": W, "C": C, "stride": s, "pad": p, "dil": d,
                    "Ho": Ho, "Wo": Wo, "read_bytes": read, "write_bytes": write, "fn": fn, "args": args})
    return out


def time_call(fn: Callable, args, iters: int = 20, flush: Optional[torch.Tensor] = None, cold: bool = False) -> float:
    """Median microseconds of one launch between two events on the current stream (the plan's calls carry that stream's handle)."""
    ts = []
    for _ in range(iters):
        if cold:
            flush.zero_()
        else:
            assert fn(*args) == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        assert rc == 0, getattr(fn, "__name__", fn)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def depthwise_table(plan, iters: int = 20, with_cold: bool = True, yardsticks: bool = True) -> List[Dict]:
    """depthwise_calls() + us warm / cold, TB/s against the algorithmic bytes, and the same traffic through trivial kernels: a torch
    copy with the same read + written bytes (fwd / bwd-data), pp_yardstick_stream_read over the same read bytes (bwd-weight)."""
    from . import _lib
    L = _lib.lib_real()
    dev = torch.device("cuda", torch.cuda.current_device())
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev) if with_cold else None
    sink = torch.zeros(4, device=dev)
    rows = []
    for c in depthwise_calls(plan):
        fn, args = c.pop("fn"), c.pop("args")
        tot = c["read_bytes"] + c["write_bytes"]
        c["us_warm"] = round(time_call(fn, args, iters), 2)
        c["TBps_warm"] = round(tot / c["us_warm"] / 1e6, 3)
        if with_cold:
            c["us_cold"] = round(time_call(fn, args, max(iters // 2, 5), flush, True), 2)
            c["TBps_cold"] = round(tot / c["us_cold"] / 1e6, 3)
        if yardsticks:
            if c["kind"] == "bwd_weight":
                buf = torch.empty(c["read_bytes"] // 4, device=dev).normal_()
                st = torch.cuda.current_stream().cuda_stream

                def yfn(_buf=buf, _st=st):
                    return L.pp_yardstick_stream_read(_buf.data_ptr(), _buf.numel() * 4, -1024, sink.data_ptr(), _st)
                c["yardstick"] = "read-only kernel over x + dy bytes"
            else:
                src = torch.empty(c["read_bytes"] // 4, device=dev).normal_()
                dst = torch.empty(c["write_bytes"] // 4, device=dev)
                n = min(src.numel(), dst.numel())
                # read `read_bytes`, write `write_bytes`: a strided gather for the stride-2 layers would not be a trivial kernel, so the
                # yardstick copies the smaller extent and reads the rest
                extra = src.numel() - n

                def yfn(_s=src, _d=dst, _n=n, _e=extra):
                    _d[:_n].copy_(_s[:_n])
                    if _e > 0:
                        L.pp_yardstick_stream_read(_s[_n:].data_ptr(), _e * 4 // 16 * 16, -256, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
                    return 0
                c["yardstick"] = "torch copy (+ read-only kernel for the unread rest)" if extra > 0 else "torch copy of the same bytes"
            c["yard_us_warm"] = round(time_call(yfn, (), iters), 2)
            if with_cold:
                c["yard_us_cold"] = round(time_call(yfn, (), max(iters // 2, 5), flush, True), 2)
            c["x_yardstick_warm"] = round(c["us_warm"] / c["yard_us_warm"], 2)
        rows.append(c)
    return rows


def depthwise_summary(rows: List[Dict], large_bytes: int = 16 << 20) -> Dict:
    """Aggregate TB/s per kind over the step's depthwise launches and the slowest large (>= 16 MB moved) layer of each kind."""
    out = {}
    for kind in ("fwd", "bwd_data", "bwd_weight"):
        rs = [r for r in rows if r["kind"] == kind]
        if not rs:
            continue
        by = sum(r["read_bytes"] + r["write_bytes"] for r in rs)
        us = sum(r["us_warm"] for r in rs)
        big = [r for r in rs if r["read_bytes"] + r["write_bytes"] >= large_bytes]
        worst = min(big, key=lambda r: r["TBps_warm"]) if big else None
        out[kind] = {"launches": len(rs), "bytes_per_step": by, "us_per_step_warm": round(us, 1), "TBps_warm": round(by / us / 1e6, 3),
                     "frac_of_8TBps": round(by / us / 1e6 / 8.0, 4),
                     "large_layers_TBps_warm": [r["TBps_warm"] for r in big],
                     "worst_large_layer": ({k: worst[k] for k in ("entry", "H", "W", "C", "stride", "us_warm", "TBps_warm")} if worst else None)}
    return out
