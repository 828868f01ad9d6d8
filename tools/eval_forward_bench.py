#!/usr/bin/env python3
"""Inference forward of DeepLabv3+-MobileNetV2 (the network part of an acquisition / validation round): images/s at a few
batch sizes, low-resolution-logits form (DeepLab.forward_lowres) and full form (model(x)["pred"])."""
import os, sys, time, warnings
from argparse import Namespace
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd.utils.utils import get_model

def main():
    C = 19
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name="deeplab", weight_type="random")).cuda().eval()
    for B in [int(v) for v in sys.argv[1:]] or [1, 4, 16]:
        x = torch.randn(B, 3, 256, 512, device="cuda")
        with torch.no_grad():
            for name, fn in (("lowres", lambda: m.forward_lowres(x)), ("full", lambda: m(x)["pred"])):
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                n = 30
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                print(f"B={B:3d} {name:6s}: {dt * 1e3:7.3f} ms/forward  {B / dt:8.1f} images/s")

if __name__ == "__main__":
    main()
