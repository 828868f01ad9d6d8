#!/usr/bin/env python3
"""TIMING ONLY: what would it buy to take the two SegmentHead weight gradients (1.05 ms of MFMA work) out of the contended
backward window?  mode 1: skip them (upper bound).  mode 2: run the weight gradients of step t on the side stream at the START
of step t+1, under its encoder forward (their result is not applied - numerics are wrong, the overlap is real).  mode 3: run them
when the backward reaches the encoder (under the encoder's latency-bound backward instead of beside the head's own backward-data and
BatchNorm launches).  mode 4: run both once the first head convolution's backward-data is enqueued."""
import os, sys, time, warnings
from argparse import Namespace
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import engine as E, _lib
from pixelpick_amd.trainer import FlatTrainer
from pixelpick_amd.utils.utils import get_model
from bench import synth_train_batch
warnings.simplefilter("ignore")
torch.manual_seed(0)
m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=19, network_name="deeplab", weight_type="random")).cuda().train()
tr = FlatTrainer(m, ignore_index=19)
x, y = synth_train_batch(4, 19, 256, 512, 20, torch.device("cuda"), 1)
orig = E._conv2d_bwd
mode = [0]
stash = []
L = _lib.lib()
def is_head(w, dil): return w.shape[0] == 3 and w.shape[2] >= 256 and w.shape[3] == 256 and dil == 1
def patched(tape, dy, x_, w, bias, stride, pad, dil):
    if mode[0] and is_head(w, dil):
        if mode[0] >= 2:
            stash.append((x_.t, dy, w, stride, pad, dil))
        rg = w.requires_grad
        w.requires_grad_(False)
        try:
            r = orig(tape, dy, x_, w, bias, stride, pad, dil)
        finally:
            w.requires_grad_(rg)
        if mode[0] == 4 and w.shape[2] == 304:
            launch_stashed(tape)
        return r
    return orig(tape, dy, x_, w, bias, stride, pad, dil)
E._conv2d_bwd = patched
_tape_init = E.Tape.__init__
def tape_init(self, enabled=True):
    _tape_init(self, enabled)
    if mode[0] == 3:
        self.hooks["encoder_done"] = launch_stashed
E.Tape.__init__ = tape_init
side = E._side_stream(torch.device("cuda:0"), 0)
dwbuf = {}
def launch_stashed(tape=None):
    if not stash: return
    ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream()); side.wait_event(ev)
    if tape is not None:
        if tape._side is None: tape._side = {}
        tape._side[0] = side                      # joined at the end of backward
    for (xt, dy, w, stride, pad, dil) in stash:
        B, H, W, Cin, ldx = E._geom(xt); _, Ho, Wo, Cout, lddy = E._geom(dy)
        kh, kw = w.shape[0], w.shape[1]
        n = int(L.pp_conv2d_bwd_weight_workspace_bytes(B, H, W, Cin, Cout, kh, kw, stride, pad, dil))
        key = (Cin, Cout)
        if key not in dwbuf: dwbuf[key] = (torch.empty_like(w), torch.empty(n, dtype=torch.uint8, device="cuda"))
        dw, ws = dwbuf[key]
        xt.record_stream(side); dy.record_stream(side)
        _lib.check(L.pp_conv2d_bwd_weight(xt.data_ptr(), ldx, B, H, W, Cin, dy.data_ptr(), lddy, Cout, kh, kw, stride, pad, dil, dw.data_ptr(), None, ws.data_ptr(), ws.numel(), side.cuda_stream), "wgrad")
    stash.clear()
def run(n=30):
    for _ in range(5):
        if mode[0] == 2: launch_stashed()
        tr.train_step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        if mode[0] == 2: launch_stashed()
        tr.train_step(x, y)
    if mode[0] == 2: launch_stashed()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for md in (0, 1, 3, 4, 0, 3, 4, 2):
    mode[0] = md
    stash.clear()
    print({0: "full step                       ", 1: "head wgrads skipped (bound)     ", 2: "head wgrads under the next fwd  ",
           3: "head wgrads under the encoder bwd", 4: "head wgrads after conv1 bwd-data "}[md], round(run(), 3), "ms")
