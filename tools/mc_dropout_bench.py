#!/usr/bin/env python3
"""MC-dropout acquisition round (query.py:176-188 as intended: mean over mc_n_steps stochastic passes) through the real
DeepLabv3+-MobileNetV2: all passes of an image in one forward (mc_chunk = 32) vs one forward per pass (mc_chunk = 1)."""
import io, contextlib, os, sys, tempfile, time, warnings
from argparse import Namespace
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd import query as ppq
from pixelpick_amd.utils.utils import get_model
warnings.simplefilter("ignore")
C, h, w, n = 19, 256, 512, 16
m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name="deeplab", weight_type="random")).cuda()


class DS:
    def __init__(s):
        s.xs = torch.randn(n, 3, h, w); s.ys = torch.randint(0, C, (n, h, w)); s.queries = [np.zeros((h, w), bool) for _ in range(n)]

    def label_queries(s, d, k):
        pass


class DL:
    def __init__(s, d):
        s.dataset = d

    def __iter__(s):
        for i in range(n):
            yield {"x": s.dataset.xs[i][None], "y": s.dataset.ys[i][None], "p_img": [f"/i{i}.png"]}


with tempfile.TemporaryDirectory() as td:
    a = Namespace(dataset_name="cs", debug=False, dir_root=td, experim_name="mc", ignore_index=C, mc_n_steps=20, n_classes=C,
                  n_pixels_by_us=20, network_name="deeplab", weight_type="random", query_strategy="entropy", reverse_order=False, stride_total=16,
                  top_n_percent=0.0, use_mc_dropout=True, vote_type="hard")
    for chunk in (32, 1):
        a.mc_chunk = chunk
        qs = ppq.QuerySelector(a, DL(DS()), device=torch.device("cuda:0"))
        with contextlib.redirect_stdout(io.StringIO()):
            qs(1, m)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            qs(2, m)
        torch.cuda.synchronize()
        print(f"mc_n_steps=20, mc_chunk={chunk:2d}: {n / (time.perf_counter() - t0):6.1f} images/s")
