"""Pins oracle/net.py to the imported reference LAYER BY LAYER (CPU): every Conv2d / BatchNorm2d / GroupNorm output, the
gradient arriving at it and every parameter gradient of one train step, against tests/golden/trace_*.npz
(tools/gen_golden_net_trace.py).  These are the tensors tests/test_layerwise_parity_gpu.py forces the HIP network with,
so the chain is: reference == oracle (here, per module) and oracle == HIP kernels (there, per module, on the GPU).

Both sides are torch CPU ops in the same order, evaluated single-threaded, so the agreement is expected to be exact; the
bar is 1e-5 of each tensor's scale (no noise term)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import formula_init as fi
from oracle.net import OracleDeepLab, OracleFPN
from trace_summary import trace_model

TOL = 1e-5


@pytest.mark.parametrize("network", ["deeplab", "fpn"])
def test_oracle_trace_matches_reference_module_by_module(golden_dir, network):
    g = np.load(os.path.join(golden_dir, f"trace_{network}_cs64x96.npz"))
    B, H, W, C, ign, n_lab = [int(v) for v in g["shape"]]
    prev = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        o = OracleDeepLab(C, 0.0, 0.0, 0.0) if network == "deeplab" else OracleFPN(C)
        o.load_state_dict(fi.formula_state_dict(o.state_dict()))
        o.train()
        x = fi.formula_input(B, H, W, key="xcs64x96")
        y = fi.formula_labels(B, H, W, C, ign, n_lab, key="ycs64x96")
        t = trace_model(o, lambda: F.cross_entropy(o(x), y, ignore_index=ign))
    finally:
        torch.set_num_threads(prev)
    assert abs(t["loss"] - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    # same modules (the oracle may execute independent branches in another order: compare by name)
    assert sorted(t["names"].tolist()) == sorted(g["names"].tolist())
    assert t["param_names"].tolist() == g["param_names"].tolist()
    idx = {n: i for i, n in enumerate(t["names"].tolist())}
    worst = 0.0
    for what in ("fwd", "grad"):
        for j, n in enumerate(g["names"].tolist()):
            ref, got = g[what][j], t[what][idx[n]]
            scale = max(ref[2], 1e-30)                                   # abs-max of the tensor
            err = max(abs(got[1] - ref[1]) / max(ref[1], 1e-30),          # abs-sum
                      abs(got[0] - ref[0]) / max(ref[1], 1e-30),          # sum, relative to abs-sum
                      abs(got[2] - ref[2]) / scale, np.abs(got[3:] - ref[3:]).max() / scale)
            worst = max(worst, err)
            assert err <= TOL, f"{what} {n}: {err:.3e}"
    for j, n in enumerate(g["param_names"].tolist()):
        ref, got = g["param_grad"][j], t["param_grad"][j]
        scale = max(ref[2], 1e-30)
        err = max(abs(got[1] - ref[1]) / max(ref[1], 1e-30), abs(got[0] - ref[0]) / max(ref[1], 1e-30),
                  abs(got[2] - ref[2]) / scale, np.abs(got[3:] - ref[3:]).max() / scale)
        worst = max(worst, err)
        assert err <= TOL, f"param_grad {n}: {err:.3e}"
    print(f"[{network}] worst deviation from the reference trace: {worst:.2e}")
