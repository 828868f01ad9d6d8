#!/usr/bin/env python3
"""What one recorded call costs the host inside pp_plan_replay: 2000 x pp_add2d on 64 floats (the smallest entry point: one argument
check + one hipLaunchKernel) on one stream, native loop vs the same calls from Python.  GPU box: python tools/plan_overhead.py"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import _lib

L = _lib.lib()
a = torch.zeros(64, device="cuda"); b = torch.ones(64, device="cuda"); o = torch.empty(64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
N = 2000
with _lib.record_plan(native=True) as plan:
    for _ in range(N):
        _lib.check(_lib.lib().pp_add2d(a.data_ptr(), 64, b.data_ptr(), 64, o.data_ptr(), 64, 1, 64, st), "add")
torch.cuda.synchronize()
for name, fn in (("native pp_plan_replay", plan.replay), ("python list", plan.replay_python)):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t = time.perf_counter(); fn(); dt = time.perf_counter() - t
        torch.cuda.synchronize()
        best = min(best, dt)
    print(f"{name:24s} {best / N * 1e6:6.2f} us per recorded call (issued into an empty queue, {N} calls)")
