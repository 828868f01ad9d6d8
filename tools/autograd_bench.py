#!/usr/bin/env python3
"""The reference's own training idiom (model.py:113-122: model(x)['pred'] -> F.cross_entropy -> loss.backward() ->
torch.optim.Adam.step()) on the HIP networks through torch.autograd, next to FlatTrainer.train_step."""
import os, sys, time, warnings
from argparse import Namespace
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd.utils.utils import get_model, get_optimizer
from bench import synth_train_batch
warnings.simplefilter("ignore")
C = 19
args = Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name="deeplab", weight_type="random", dataset_name="cs",
                 optimizer_params={"lr": 5e-4, "betas": (0.9, 0.999), "weight_decay": 2e-4, "eps": 1e-7})
model = get_model(args).cuda().train()
opt = get_optimizer(args, model)
x, y = synth_train_batch(4, C, 256, 512, 20, torch.device("cuda"), 1)


def step():
    pred = model(x)["pred"]
    loss = F.cross_entropy(pred, y, ignore_index=C)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30
print(f"autograd idiom: {dt * 1e3:.2f} ms/step = {4 / dt:.1f} images/s, loss {loss.item():.4f}")
