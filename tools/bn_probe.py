#!/usr/bin/env python3
"""Where the time of a single-launch BatchNorm forward goes: per-block wall-clock stamps (pp_debug_set_bn_probe) of the phases
entry -> statistics pass -> block reduction -> publish -> strip combine (waits for the slowest sibling) -> rows written,
for the large and the small maps of the DeepLab step, with the launch as the event pair sees it next to them."""
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
          ("1/16 18x34x960", 2448, 960), ("1/16 16x32x160", 2048, 160)]
NAMES = ["stats pass", "block tree", "publish", "combine (wait)", "apply + write"]


def run(M, C, cold, target=0):
    L.pp_debug_set_bn_target(target)
    sync, ws = E._bn_exchange(dev)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(M, C, device=dev)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
    y = torch.empty_like(x)
    probe = torch.zeros(1024 * 8, dtype=torch.int64, device=dev)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    rows = []
    for it in range(12):
        if cold:
            flush.zero_()
        probe.zero_()
        torch.cuda.synchronize()
        L.pp_debug_set_bn_probe(probe.data_ptr())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(L.pp_bn_train_fwd_fused(x.data_ptr(), C, M, C, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, None, None, mean.data_ptr(),
                                           invstd.data_ptr(), None, 0, 2, 0.0, 0, None, y.data_ptr(), C, ws.data_ptr(), ws.numel(),
                                           sync.data_ptr(), sync.numel(), st), "fwd")
        b.record()
        torch.cuda.synchronize()
        L.pp_debug_set_bn_probe(None)
        p = probe.view(-1, 8).cpu()
        p = p[p[:, 0] > 0][:, :6].double()
        t0 = p[:, 0].min()
        ph = (p - t0) * 0.01                          # us since the first block's entry
        if it >= 2:
            rows.append((a.elapsed_time(b) * 1e3, p.shape[0], ph))
    ev = sorted(r[0] for r in rows)[len(rows) // 2]
    nb = rows[0][1]
    ph = torch.stack([r[2] for r in rows]).mean(0)    # [blocks][6]
    line = f"    blocks {nb:4d}  event pair {ev:6.1f} us | entry spread {ph[:, 0].max():5.1f} | "
    for i, n in enumerate(NAMES):
        d = ph[:, i + 1] - ph[:, i]
        line += f"{n} {d.mean():5.1f} (max {d.max():5.1f}) | "
    line += f"last block done {ph[:, 5].max():5.1f}"
    print(line)


def main():
    for name, M, C in SHAPES:
        print(f"{name}  ({M * C * 4 / 1e6:.1f} MB)")
        for cold, target in ((True, 0), (False, 0), (True, 512), (False, 512)):
            print(f"  {'cold (MALL flushed)' if cold else 'warm'}, target blocks {target or 'default (one per CU)'}")
            run(M, C, cold, target)
    L.pp_debug_set_bn_target(0)


if __name__ == "__main__":
    main()
