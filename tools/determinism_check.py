#!/usr/bin/env python3
"""Run the BASELINE train step N times from identical state, several times over, and compare bitwise."""
import os, sys, warnings, hashlib
from argparse import Namespace
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import engine as E
from pixelpick_amd.utils.utils import get_model
from pixelpick_amd.trainer import FlatTrainer
from bench import synth_train_batch
warnings.simplefilter("ignore")
C = 19
STEPS, RUNS = int(os.environ.get("STEPS", 12)), int(os.environ.get("RUNS", 5))
x, y = synth_train_batch(4, C, 256, 512, 20, torch.device("cuda"), 1)
sigs = []
for r in range(RUNS):
    torch.manual_seed(0)
    m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name="deeplab", weight_type="random")).cuda().train()
    tr = FlatTrainer(m, ignore_index=C)
    E.set_dropout_seed(1234)
    losses = []
    for s in range(STEPS):
        losses.append(tr.train_step(x, y))
    torch.cuda.synchronize()
    sig = hashlib.sha1(tr.flat_p.cpu().numpy().tobytes()).hexdigest()[:12]
    ls = [float(l) for l in losses]
    sigs.append((sig, ls))
    first_div = next((i for i, (a, b) in enumerate(zip(ls, sigs[0][1])) if a != b), None)
    print(f"run {r}: params {sig} loss[-1] {ls[-1]:.9f} first diverging step vs run 0: {first_div}")
print("DETERMINISTIC" if len({s for s, _ in sigs}) == 1 else "NON-DETERMINISTIC")
