#!/usr/bin/env python3
"""Op-level timing of the reference's default top-5 % mode (args.py:25: k = 5 % of the pixels) on BASELINE configs[1]:
pp_acq_score_topk at B=256 x 256x512x19, entropy, k = 6553.  Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys
os.environ.setdefault("PIXELPICK_KNOBS_BUILD", "1")      # the pp_debug_* planner switches live in the test build only
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import acquisition as acq, _lib
B, C, H, W = int(os.environ.get("B", 256)), 19, 256, 512
k = int(os.environ.get("K", H * W * 5 // 100))
torch.manual_seed(0)
x = torch.randn(B, C, H, W, device="cuda") * 3
L = _lib.lib()
idx = torch.empty((B, k), dtype=torch.int32, device="cuda"); val = torch.empty((B, k), device="cuda")
ws = torch.empty(int(L.pp_acq_workspace_bytes(B, C, H, W, k)), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
L.pp_debug_set_acq_tuning(int(os.environ.get("TUNE", "0")), 0); L.pp_debug_set_reduce_mode(int(os.environ.get("RMODE", "0")))
excl = (torch.rand(B, H, W, device="cuda") < 0.05).to(torch.uint8) if os.environ.get("MASK") else None     # MASK=1: bench.py's 5 % exclusion mask
def op():
    _lib.check(L.pp_acq_score_topk(x.data_ptr(), B, C, H, W, *x.stride(), excl.data_ptr() if excl is not None else None, 0, k, idx.data_ptr(), val.data_ptr(), None, ws.data_ptr(), ws.numel(), st), "op")
for _ in range(5): op()
ts = []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); op(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort()
t = ts[len(ts) // 2]
gb = B * H * W * (C * 4 + 1) / 1e9
print(f"k={k}: op {t * 1e3:.1f} us (median of 30; min {ts[0] * 1e3:.1f}) = {gb / (t * 1e-3):.1f} GB/s = {gb / (t * 1e-3) / 8000:.3f} of 8 TB/s; with the map write counted {(gb + B * H * W * 4 / 1e9) / (t * 1e-3) / 8000:.3f}")
# picks = the k best of the device's own map
L.pp_debug_set_acq_tuning(0, 0); L.pp_debug_set_reduce_mode(0)
i2, v2, m = acq.score_topk(x[:2], excl[:2].bool() if excl is not None else None, "entropy", k, return_map=True)
ref = torch.sort(m[0].reshape(-1), descending=True, stable=True)
assert torch.equal(ref.indices[:k].to(torch.int32), i2[0]), "picks differ from a stable sort of the map"
print("picks == stable descending sort of the map")
