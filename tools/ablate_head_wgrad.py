import os, sys, time, json, warnings
from argparse import Namespace
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import engine as E, _lib
from pixelpick_amd.trainer import FlatTrainer
from pixelpick_amd.utils.utils import get_model
from bench import synth_train_batch
warnings.simplefilter("ignore")
torch.manual_seed(0)
m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=19, network_name="deeplab")).cuda().train()
tr = FlatTrainer(m, ignore_index=19)
x, y = synth_train_batch(4, 19, 256, 512, 20, torch.device("cuda"), 1)
orig = E._conv2d_bwd
mode = [0]
def patched(tape, dy, x_, w, bias, stride, pad, dil):
    if mode[0] and w.shape[0] == 3 and w.shape[2] >= 256 and w.shape[3] == 256 and dil == 1:
        # TIMING ONLY: no weight gradient for the two SegmentHead convolutions
        rg = w.requires_grad
        w.requires_grad_(False)
        try:
            return orig(tape, dy, x_, w, bias, stride, pad, dil)
        finally:
            w.requires_grad_(rg)
    return orig(tape, dy, x_, w, bias, stride, pad, dil)
E._conv2d_bwd = patched
def run(n=20):
    for _ in range(5): tr.train_step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.train_step(x, y)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for md in (0, 1, 0, 1):
    mode[0] = md
    print("skip head wgrad" if md else "full step      ", round(run(), 3), "ms")
