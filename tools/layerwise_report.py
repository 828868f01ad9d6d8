#!/usr/bin/env python3
"""Per-layer parity tables of the HIP networks against the plain-PyTorch oracle (tests/layerwise.py harness):
forced / branch-forced / free-running, oracle evaluated in fp32 or fp64.  Runs on the GPU box.

    python tools/layerwise_report.py deeplab 4 128 192 [out.txt]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.nn.functional as F  # noqa: E402
import formula_init as fi  # noqa: E402
import test_layerwise_parity_gpu as T  # noqa: E402
from layerwise import LayerwiseParity, OracleTrace  # noqa: E402
from pixelpick_amd.trainer import FlatTrainer  # noqa: E402


def run(network, B, H, W, force, dtype, key="free"):
    m, o = T._models(network, 19)
    o = o.to(dtype)
    x = fi.formula_input(B, H, W, key=f"x{key}")
    y = fi.formula_labels(B, H, W, 19, 19, 20, key=f"y{key}")
    tr = OracleTrace(o)
    loss = F.cross_entropy(o(x.to(dtype)), y, ignore_index=19)
    loss.backward()
    tr.close()
    for d in (tr.fwd, tr.grad):
        for k in d:
            d[k] = d[k].float()
    ograds = {n: p.grad.float() for n, p in o.named_parameters()}
    t = FlatTrainer(m, ignore_index=19)
    with LayerwiseParity(m, tr, force=force) as lp:
        t.forward_backward(x.cuda(), y.cuda())
        lp.compare_param_grads({n: t._grad_view[id(p)] for n, p in m.named_parameters()}, ograds)
    return lp


def main():
    network, B, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    out = open(sys.argv[5], "w") if len(sys.argv) > 5 else sys.stdout
    for force, dtype in ((True, torch.float32), ("branch", torch.float32), ("branch", torch.float64), (False, torch.float32)):
        lp = run(network, B, H, W, force, dtype)
        tag = f"{network} B={B} {H}x{W} force={force} oracle={str(dtype).split('.')[-1]}"
        print(f"==== {tag}", file=out)
        print(lp.summary(), file=out)
        if force == "branch":
            print("-- arriving-gradient rel-L2 in backward order / parameter gradients by layer", file=out)
            pg = {}
            for k, n, e, _ in lp.rec:
                if k == "param_grad":
                    pg.setdefault(n.rpartition(".")[0], []).append(f"{n.rpartition('.')[2]}={e:.1e}")
            for k, n, e, _ in lp.rec:
                if k == "dy":
                    print(f"   {n:46s} dy {e:.2e}   flips {lp.flips.get(n, ('-', 0))[0]}   {' '.join(pg.get(n, []))}", file=out)
    out.flush()


if __name__ == "__main__":
    main()
