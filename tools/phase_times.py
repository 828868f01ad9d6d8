#!/usr/bin/env python3
"""Where one DeepLabv3+-MNv2 train step spends its GPU time, from HIP events on the main stream (no profiler):
forward, backward main chain, wait for the side-stream weight gradients (join), optimiser."""
import os, sys, warnings
from argparse import Namespace
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import engine as E
from pixelpick_amd.utils.utils import get_model
from pixelpick_amd.trainer import FlatTrainer
from bench import synth_train_batch
warnings.simplefilter("ignore")
m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=19, network_name="deeplab", weight_type="random")).cuda().train()
tr = FlatTrainer(m, ignore_index=19)
x, y = synth_train_batch(4, 19, 256, 512, 20, torch.device("cuda"), 1)
for _ in range(5):
    tr.train_step(x, y)
torch.cuda.synchronize()
N = 20
acc = {}
for overlap in (True, False):
    E.Tape.overlap_wgrad = overlap
    for _ in range(3):
        tr.train_step(x, y)
    tot = {}
    for _ in range(N):
        E.Tape.trace = []
        s = E._mark()
        tr.train_step(x, y)
        e = E._mark()
        torch.cuda.synchronize()
        marks = [("start", s)] + E.Tape.trace + [("end", e)]
        for (la, ea), (lb, eb) in zip(marks, marks[1:]):
            tot[f"{la}->{lb}"] = tot.get(f"{la}->{lb}", 0.0) + ea.elapsed_time(eb)
    E.Tape.trace = None
    print("overlap_wgrad =", overlap, {k: round(v / N, 3) for k, v in tot.items()}, "sum", round(sum(tot.values()) / N, 3))
