#!/usr/bin/env python3
"""Random shapes / class counts / k / strategies / exclusion fractions / map kinds (i.i.d., smooth, blocky, quantised) and sample settings:
the list select (pp_acq_score_topk without a caller's map, k > 48) must return the picks and values of the map path
(pp_debug_set_reduce_mode bit 11).  GPU box:  python tools/probe/fuzz_list_select.py   (120 cases, ~2 s; round 6: 0 mismatches)"""
import os, sys, random
os.environ["PIXELPICK_KNOBS_BUILD"] = "1"
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pixelpick_amd import _lib, acquisition as acq
L = _lib.lib()
random.seed(1)
gen = torch.Generator(device="cuda").manual_seed(1)
STR = ["entropy", "least_confidence", "margin_sampling"]
bad = 0
for it in range(120):
    C = random.choice([11, 19, 21])
    B = random.randint(1, 6)
    H = random.randint(40, 300); W = random.choice([64, 96, 128, 172, 256, 320, 500, 512])
    while H * W < 16384 or (H * W) % 4: H += 1
    N = H * W
    k = random.randint(49, min(N // 8, 7281))
    st = random.choice(STR)
    scale = random.choice([0.05, 0.5, 3.0, 10.0])
    kind = random.choice(["iid", "smooth", "blocky", "quant"])
    if kind == "iid":
        x = torch.randn((B, C, H, W), device="cuda", generator=gen) * scale
    elif kind == "smooth":
        x = torch.nn.functional.interpolate(torch.randn((B, C, max(H // 16, 2), max(W // 16, 2)), device="cuda", generator=gen) * scale, size=(H, W), mode="bilinear").contiguous()
    elif kind == "blocky":
        x = torch.randn((B, C, H, W), device="cuda", generator=gen) * 0.01
        x[:, :, : H // 3] += torch.randn((B, C, 1, 1), device="cuda", generator=gen) * scale       # a confident third, an uncertain rest
    else:
        x = torch.round(torch.randn((B, C, H, W), device="cuda", generator=gen) * 2) / 2
    ex = None
    if random.random() < 0.7:
        ex = torch.rand((B, H, W), device="cuda", generator=gen) < random.choice([0.02, 0.3, 0.9])
    mode = random.choice([0, 0, 0, 1 << 18, (2 << 18) | (2 << 20), 24 << 12, 63 << 12])
    try:
        L.pp_debug_set_reduce_mode(2048)
        ref = acq.score_topk(x, ex, st, k)
        L.pp_debug_set_reduce_mode(mode)
        got = acq.score_topk(x, ex, st, k)
    finally:
        L.pp_debug_set_reduce_mode(0)
    ok = torch.equal(ref[0], got[0]) and torch.equal(torch.nan_to_num(ref[1], nan=-7.0), torch.nan_to_num(got[1], nan=-7.0))
    if not ok:
        bad += 1
        print("MISMATCH", it, C, B, H, W, k, st, kind, scale, mode)
print("fuzz: 120 cases,", bad, "mismatches")
