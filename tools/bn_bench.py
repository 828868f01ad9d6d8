#!/usr/bin/env python3
"""Single-launch BatchNorm forward / backward on the DeepLab train-step shapes in isolation: time per launch and the
HBM rate against the algorithmic bytes (fwd: read x, write y = 2S - the second read of x is an L2/MALL hit when it fits;
bwd: read x, dy, write dx = 3S).  `python tools/bn_bench.py [bytes_per_block ...]` sweeps the large-map grid knob."""
import os
os.environ.setdefault("PIXELPICK_KNOBS_BUILD", "1")      # the pp_debug_* planner switches live in the test build only
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import _lib  # noqa: E402
from pixelpick_amd import engine as E  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
SHAPES = [("head 64x128x256", 4 * 64 * 128, 256), ("block2 expand 130x258x96", 4 * 130 * 258, 96), ("stem 128x256x32", 4 * 128 * 256, 32),
          ("block3 expand 66x130x144", 4 * 66 * 130, 144), ("low-level 64x128x48", 4 * 64 * 128, 48), ("1/16 18x34x960", 2448, 960),
          ("1/16 16x32x160", 2048, 160)]


def bench(M, C, iters=30):
    sync, ws = E._bn_exchange(dev)
    st = torch.cuda.current_stream().cuda_stream
    x, dy = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    mean, invstd, dg, db = (torch.empty(C, device=dev) for _ in range(4))
    y, dx = torch.empty_like(x), torch.empty_like(x)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)

    def fwd():
        _lib.check(L.pp_bn_train_fwd_fused(x.data_ptr(), C, M, C, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, None, None, mean.data_ptr(),
                                           invstd.data_ptr(), None, 0, 2, 0.0, 0, None, y.data_ptr(), C, ws.data_ptr(), ws.numel(),
                                           sync.data_ptr(), sync.numel(), st), "fwd")

    def bwd():
        _lib.check(L.pp_bn_bwd_fused(x.data_ptr(), C, dy.data_ptr(), C, None, C, 2, M, C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                     dg.data_ptr(), db.data_ptr(), dx.data_ptr(), C, None, 0, 1.0, beta.data_ptr(), ws.data_ptr(), ws.numel(),
                                     sync.data_ptr(), sync.numel(), st), "bwd")
    # yardsticks with the same traffic through trivial elementwise launches: y = x (read S, write S) and dx = x + dy (read 2S, write S)
    def copy():
        y.copy_(x)

    def add():
        torch.add(x, dy, out=dx)
    out = []
    for fn in (fwd, bwd, copy, add):
        fn()
        for cold in (True, False):
            ts = []
            for _ in range(iters):
                if cold:
                    flush.zero_()                   # evict x from the 256 MiB MALL: the launch starts cold
                else:
                    fn()                            # warm: the tensors were touched by the launch before (in the step the producer has just written x)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            ts.sort()
            out.append(ts[len(ts) // 2])
    return out


def main():
    knobs = [int(v) for v in sys.argv[1:]] or [48 << 10]
    print("capacity (blocks):", L.pp_bn_fused_capacity())
    for kb in knobs:
        L.pp_debug_set_bn_bytes_per_block(kb)
        print(f"== bytes per block {kb}")
        print("  map                            bytes S | forward: cold, warm us (TB/s of 2S) | y = x: cold, warm us | backward: cold, warm us (TB/s of 3S) | dx = x + dy: cold, warm us")
        for name, M, C in SHAPES:
            fc, fw, bc, bw, cc, cw, ac, aw = bench(M, C)
            S = M * C * 4
            print(f"  {name:28s} {S / 1e6:6.1f} MB | {fc:6.1f} {fw:6.1f} ({2 * S / fc / 1e6:4.2f} {2 * S / fw / 1e6:4.2f}) | {cc:6.1f} {cw:6.1f} ({2 * S / cc / 1e6:4.2f} {2 * S / cw / 1e6:4.2f}) |"
                  f" {bc:6.1f} {bw:6.1f} ({3 * S / bc / 1e6:4.2f} {3 * S / bw / 1e6:4.2f}) | {ac:6.1f} {aw:6.1f} ({3 * S / ac / 1e6:4.2f} {3 * S / aw / 1e6:4.2f})")


if __name__ == "__main__":
    main()
