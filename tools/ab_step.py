#!/usr/bin/env python3
"""In-process A/B of the BASELINE train step for a debug knob of the library: alternates the settings on one model /
one box (box-to-box variance is +-0.1 ms, more than most single optimisations).
  python tools/ab_step.py pp_debug_set_dw_variant 0 1
  python tools/ab_step.py pp_debug_set_conv_variant 0 64"""
import os, sys, time, warnings
os.environ.setdefault("PIXELPICK_KNOBS_BUILD", "1")      # the pp_debug_* planner switches live in the test build only
from argparse import Namespace
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import _lib
from pixelpick_amd.utils.utils import get_model
from pixelpick_amd.trainer import FlatTrainer
from bench import synth_train_batch
warnings.simplefilter("ignore")
fn_name, values = sys.argv[1], [int(v, 0) for v in sys.argv[2:]]
L = _lib.lib()
setter = getattr(L, fn_name)
NET = os.environ.get("NET", "deeplab")
m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=19, network_name=NET, weight_type="random", n_layers=50,
                        use_softmax=True, use_dilated_resnet=True, width_multiplier=1.0)).cuda().train()
tr = FlatTrainer(m, ignore_index=19)
x, y = synth_train_batch(4, 19, 256, 512, 20, torch.device("cuda"), 1)
for _ in range(5):
    tr.train_step(x, y)
for rep in range(3):
    for v in values:
        setter(v)
        for _ in range(3):
            tr.train_step(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N = 40 if NET == "deeplab" else 12
        for _ in range(N):
            tr.train_step(x, y)
        torch.cuda.synchronize()
        print(f"{fn_name}({v}): {(time.perf_counter() - t0) / N * 1e3:.3f} ms/step")
setter(values[0])
