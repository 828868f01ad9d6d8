#!/usr/bin/env python3
"""How far do the HIP network's train-mode activations sit from a plain-PyTorch evaluation, layer by layer?

Runs on the GPU box (needs no reference): the HIP DeepLab and oracle/net.py's OracleDeepLab get the same formula
weights; every BatchNorm output (after its activation) is compared with the oracle's fp32 AND fp64 evaluation:
max |delta| / std(layer) over units active in both, and the number of ReLU/ReLU6 units whose branch differs.
Sizes the threshold margin of the well-conditioned gradient fixtures (tools/gen_golden_net_tight.py).

    python tools/act_deviation.py [B H W]
"""
import os
import sys
import warnings
from argparse import Namespace

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import formula_init as fi  # noqa: E402
from oracle.net import OracleDeepLab  # noqa: E402
from pixelpick_amd import engine as E  # noqa: E402
from pixelpick_amd.networks import layers as L  # noqa: E402
from pixelpick_amd.utils.utils import get_model  # noqa: E402


def main():
    B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (2, 64, 96)
    C = 19
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name="deeplab", weight_type="random"))
    sd = fi.formula_state_dict(m.state_dict())
    m.load_state_dict(sd)
    for mod in m.modules():
        if isinstance(mod, L.Dropout):
            mod.p = 0.0
    m = m.cuda().train()
    names = {id(mod): n for n, mod in m.named_modules()}
    ours = {}
    orig_run = L.BatchNorm2d.run

    def run(self, tape, x, act=E.ACT_NONE, residual=None, dst=None, dropout=None):
        out = orig_run(self, tape, x, act, residual, dst, dropout)
        ours[names[id(self)]] = (act, out.t.detach().permute(0, 3, 1, 2).cpu().double())
        return out

    L.BatchNorm2d.run = run
    x = fi.formula_input(B, H, W, key="dev")
    with torch.no_grad():
        m(x.cuda())
    L.BatchNorm2d.run = orig_run

    res = {}
    for dt in (torch.float32, torch.float64):
        o = OracleDeepLab(C, 0.0, 0.0, 0.0)
        o.load_state_dict(sd)
        o = o.to(dt).train()
        got = {}
        hooks = [mod.register_forward_hook(lambda mod, i, out, n=n: got.__setitem__(n, out.detach().clone().double()))
                 for n, mod in o.named_modules() if isinstance(mod, torch.nn.BatchNorm2d)]
        with torch.no_grad():
            o(x.to(dt))
        for h in hooks:
            h.remove()
        res[dt] = got

    tot_units = 0
    print(f"{'layer':44s} {'units':>9s} {'dev32':>9s} {'dev64':>9s} {'flip32':>6s} {'flip64':>6s} {'min|pre|/std':>12s}")
    worst = {torch.float32: 0.0, torch.float64: 0.0}
    flips = {torch.float32: 0, torch.float64: 0}
    for n, (act, y) in ours.items():
        row = []
        for dt in (torch.float32, torch.float64):
            pre = res[dt][n]
            std = pre.std().item()
            if act == E.ACT_NONE:
                ref, on_r, on_o = pre, torch.ones_like(pre, dtype=torch.bool), torch.ones_like(pre, dtype=torch.bool)
            else:
                hi = 6.0 if act == E.ACT_RELU6 else float("inf")
                ref = pre.clamp(0.0, hi)
                on_r = (pre > 0) & (pre < hi)
                on_o = (y > 0) & (y < hi)
            both = on_r & on_o
            dev = ((y - ref).abs()[both].max().item() / std) if both.any() else 0.0
            nfl = int((on_r != on_o).sum().item()) if act != E.ACT_NONE else 0
            worst[dt] = max(worst[dt], dev)
            flips[dt] += nfl
            row += [dev, nfl]
            if dt == torch.float64 and act != E.ACT_NONE:
                marg = pre.abs().min().item() / std
                if act == E.ACT_RELU6:
                    marg = min(marg, (pre - 6.0).abs().min().item() / std)
        if act != E.ACT_NONE:
            tot_units += y.numel()
        print(f"{n:44s} {y.numel():9d} {row[0]:9.2e} {row[2]:9.2e} {row[1]:6d} {row[3]:6d} "
              f"{(marg if act != E.ACT_NONE else float('nan')):12.2e}")
    print(f"activation units: {tot_units}; worst relative deviation vs fp32 {worst[torch.float32]:.2e}, vs fp64 "
          f"{worst[torch.float64]:.2e}; flipped units vs fp32 {flips[torch.float32]}, vs fp64 {flips[torch.float64]}")
    # the oracle against itself: fp32 vs fp64
    w = 0.0
    for n in ours:
        a, b = res[torch.float32][n], res[torch.float64][n]
        w = max(w, (a - b).abs().max().item() / b.std().item())
    print(f"oracle fp32 vs fp64 worst relative pre-activation deviation: {w:.2e}")


if __name__ == "__main__":
    main()
