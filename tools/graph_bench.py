#!/usr/bin/env python3
"""Eager vs hipGraph replay of the DeepLab train step (B=4, 256x512) and, under rocprofv3 --kernel-trace, how many HIP
queues the replay actually uses.  GRAPH=1 selects the hipGraph replay, REPLAY=1 the launch-plan replay
(FlatTrainer.enable_replay); SIDE=0 records the weight gradients on the main stream."""
import json
import os
import sys
import time
import warnings
from argparse import Namespace

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import engine as E  # noqa: E402
from pixelpick_amd.trainer import FlatTrainer  # noqa: E402
from pixelpick_amd.utils.utils import get_model  # noqa: E402
from bench import synth_train_batch  # noqa: E402

warnings.simplefilter("ignore")
graph = os.environ.get("GRAPH", "0") == "1"
replay = os.environ.get("REPLAY", "0") == "1"
E.Tape.overlap_wgrad = os.environ.get("SIDE", "1") == "1"
torch.manual_seed(0)
m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=19, network_name="deeplab", weight_type="random")).cuda().train()
tr = FlatTrainer(m, ignore_index=19)
x, y = synth_train_batch(4, 19, 256, 512, 20, torch.device("cuda"), 1)
for _ in range(3):
    tr.train_step(x, y)
if graph:
    tr.enable_graph(x, y)
if replay:
    tr.enable_replay(x, y)
for _ in range(3):
    tr.train_step(x, y)
torch.cuda.synchronize()
steps = int(os.environ.get("STEPS", 50))
t0 = time.perf_counter()
for _ in range(steps):
    tr.train_step(x, y)
torch.cuda.synchronize()
t2 = time.perf_counter()
host = []
for _ in range(10):                     # host cost of issuing ONE step into an empty queue (no back-pressure from the GPU)
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    tr.train_step(x, y)
    host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
t1 = t0 + sorted(host)[len(host) // 2] * steps
print(json.dumps({"graph": graph, "launch_plan": len(tr._plan) if replay else None, "side_stream": E.Tape.overlap_wgrad, "ms_per_step": (t2 - t0) / steps * 1e3,
                  "host_enqueue_ms_per_step": (t1 - t0) / steps * 1e3}))
