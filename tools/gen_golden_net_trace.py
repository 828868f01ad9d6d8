#!/usr/bin/env python3
"""Layer-granular trace of the IMPORTED reference networks (authoring container only): for every Conv2d / BatchNorm2d /
GroupNorm module the summary (sum, abs-sum, abs-max + 8 strided samples) of its forward output and of the gradient that
arrives at that output in `F.cross_entropy(model(x)["pred"], y, ignore_index).backward()` (model.py:113-121), plus the
summary of every parameter gradient.  tests/test_oracle_trace_golden.py holds oracle/net.py to it module by module, which
pins the tensors the GPU layer-wise parity tests (tests/test_layerwise_parity_gpu.py) force the HIP network with.
Data only; single-threaded so that the summation order does not depend on the core count.
"""
import contextlib
import io
import os
import sys
from argparse import Namespace

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import networks.mobilenet_v2 as ref_mnv2  # noqa: E402
ref_mnv2.MobileNetV2._load_pretrained_model = lambda self: None
from utils.utils import get_model  # noqa: E402
import formula_init as fi  # noqa: E402
from trace_summary import trace_model  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def gen(network, tag, B, H, W, C=19, ign=19, n_lab=20):
    torch.set_num_threads(1)
    args = Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name=network, weight_type="random",
                     use_dilated_resnet=True, n_layers=50, width_multiplier=1.0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = get_model(args)
    model.load_state_dict(fi.formula_state_dict(model.state_dict()))
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    x = fi.formula_input(B, H, W, key=f"x{tag}")
    y = fi.formula_labels(B, H, W, C, ign, n_lab, key=f"y{tag}")
    out = trace_model(model, lambda: F.cross_entropy(model(x)["pred"], y, ignore_index=ign))
    out["shape"] = np.array([B, H, W, C, ign, n_lab])
    np.savez_compressed(os.path.join(OUT, f"trace_{'deeplab' if network == 'deeplab' else 'fpn'}_{tag}.npz"), **out)
    print("written", network, tag, len(out["names"]), "modules", len(out["param_names"]), "parameters")


if __name__ == "__main__":
    gen("deeplab", "cs64x96", 2, 64, 96)
    gen("FPN", "cs64x96", 2, 64, 96)
