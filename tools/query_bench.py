#!/usr/bin/env python3
"""End-to-end acquisition round (QuerySelector.__call__: eval forward + score + top-k + codec + stats) in images/s,
for the per-image loop of query.py:159 and for the batched forward (query_batch_size)."""
import os, sys, time, warnings, tempfile
from argparse import Namespace
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd.utils.utils import get_model
from pixelpick_amd import query as ppq

N, C, H, W = int(os.environ.get("N", 64)), 19, 256, 512


class DS:
    def __init__(self):
        g = torch.Generator().manual_seed(0)
        self.xs = torch.randn(N, 3, H, W, generator=g)
        self.ys = torch.randint(0, C, (N, H, W), generator=g)
        self.queries = [np.zeros((H, W), bool) for _ in range(N)]

    def label_queries(self, d, nth):
        pass


class DL:
    def __init__(self, ds):
        self.dataset = ds

    def __iter__(self):
        for i in range(N):
            yield {"x": self.dataset.xs[i][None], "y": self.dataset.ys[i][None], "p_img": [f"/img_{i}.png"]}


warnings.simplefilter("ignore")
model = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name="deeplab", weight_type="random")).cuda()
ds = DS()
for mode in ("k=20", "top5%"):
    for bs in (1, 4, 16):
        with tempfile.TemporaryDirectory() as td:
            a = Namespace(dataset_name="cs", debug=False, dir_root=td, experim_name="qb", ignore_index=C, mc_n_steps=20, n_classes=C,
                          n_pixels_by_us=20 if mode == "k=20" else 10, network_name="deeplab", weight_type="random", query_strategy="entropy",
                          reverse_order=False, stride_total=16, top_n_percent=0.0 if mode == "k=20" else 0.05,
                          use_mc_dropout=False, vote_type="hard", query_batch_size=bs)
            qs = ppq.QuerySelector(a, DL(ds), device=torch.device("cuda"))
            import io, contextlib
            with contextlib.redirect_stdout(io.StringIO()):
                qs(nth_query=1, model=model)          # warm-up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                qs(nth_query=2, model=model)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
        print(f"{mode:6s} query_batch_size={bs:2d}: {N / dt:7.1f} images/s ({dt / N * 1e3:.2f} ms/image)")
