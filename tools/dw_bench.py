#!/usr/bin/env python3
"""The depthwise 3x3 layers of MobileNetV2 (mobilenet_v2.py:38,52) in the BASELINE train step (B = 4,
256 x 512), launch by launch - forward, backward-data, backward-weight - with the arguments the step really passes (the recorded
launch plan's calls re-issued one at a time, pixelpick_amd/profiling.py): algorithmic bytes, us warm / cold, TB/s, and a trivial
kernel with the same traffic beside each.  GPU box:  python tools/dw_bench.py > profiles/r06_dw_layers.txt"""
import os
import sys
import warnings
from argparse import Namespace

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("PIXELPICK_MNV2_WEIGHTS", "random")
from pixelpick_amd import profiling  # noqa: E402
from pixelpick_amd.trainer import FlatTrainer  # noqa: E402
from pixelpick_amd.utils.utils import get_model  # noqa: E402


def main():
    B, H, W, C = int(os.environ.get("B", 4)), 256, 512, 19
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name="deeplab", weight_type="random")).cuda().train()
    tr = FlatTrainer(m, ignore_index=C)
    x = torch.randn(B, 3, H, W, device="cuda")
    y = torch.full((B, H, W), C, dtype=torch.int64, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    for b in range(B):
        idx = torch.randint(0, H * W, (20,), device="cuda", generator=g)
        y[b].view(-1)[idx] = torch.randint(0, C, (20,), device="cuda", generator=g)
    tr.enable_replay(x, y, warmup=2)
    for _ in range(3):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    rows = profiling.depthwise_table(tr._plan, iters=20)
    print(f"depthwise launches of one DeepLabv3+-MNv2 train step, B = {B}, {H} x {W}: {len(rows)} launches")
    print("  # kind        entry                               in HxWxC  s p d |  MB moved | us warm  cold | TB/s warm  cold | yardstick us warm cold | x yardstick")
    for r in rows:
        print(f"{r['index']:4d} {r['kind']:10s} {r['entry'][3:]:34s} {r['H']:4d}x{r['W']:<4d}x{r['C']:<4d} {r['stride']} {r['pad']} {r['dil']} |"
              f" {(r['read_bytes'] + r['write_bytes']) / 1e6:8.2f} | {r['us_warm']:7.1f} {r['us_cold']:6.1f} | {r['TBps_warm']:6.2f} {r['TBps_cold']:6.2f} |"
              f" {r['yard_us_warm']:8.1f} {r['yard_us_cold']:6.1f} | {r['x_yardstick_warm']:5.2f}   ({r['yardstick']})")
    import json
    print("summary:", json.dumps(profiling.depthwise_summary(rows)))


if __name__ == "__main__":
    main()
