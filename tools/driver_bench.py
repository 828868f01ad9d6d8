#!/usr/bin/env python3
"""Training-loop throughput THROUGH the active-learning driver (pixelpick_amd/model.py:_train_epoch: dataloader ->
H2D -> label sparsification -> train step -> device-side confusion matrix -> running loss) vs the bare train step."""
import os, sys, time, warnings, io, contextlib, tempfile
from argparse import Namespace
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd.model import Model
from pixelpick_amd.synthetic import SyntheticDataset
from pixelpick_amd.trainer import FlatTrainer
from pixelpick_amd.utils.utils import get_model
warnings.simplefilter("ignore")
C, H, W, N = 19, 256, 512, int(os.environ.get("N", 128))
ds = SyntheticDataset(N, H, W, C, C, n_init_pixels=20, seed=1)
ds_val = SyntheticDataset(4, H, W, C, C, seed=2)
mk = lambda d, b, sh: torch.utils.data.DataLoader(d, batch_size=b, shuffle=sh, drop_last=True)
with tempfile.TemporaryDirectory() as td:
    args = Namespace(dataset_name="cs", debug=False, dir_root=td, experim_name="drv", ignore_index=C, mc_n_steps=20, n_classes=C,
                     n_pixels_by_us=10, network_name="deeplab", weight_type="random", query_strategy="entropy", reverse_order=False, stride_total=16,
                     top_n_percent=0.0, use_mc_dropout=False, vote_type="hard", mc_dropout_p=0.2, n_init_pixels=20, max_budget=20,
                     n_epochs=1, lr_scheduler_type="Poly", replay_train_step=(None if "REPLAY" not in os.environ else bool(int(os.environ["REPLAY"]))),
                     optimizer_params={"lr": 5e-4, "betas": (0.9, 0.999), "weight_decay": 2e-4, "eps": 1e-7})
    dev = torch.device("cuda:0")
    m = Model(args, mk(ds, 4, True), mk(ds, 1, False), mk(ds_val, 1, False), device=dev)
    m._open_logs(td)
    model = get_model(args).to(dev)
    tr = FlatTrainer(model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=2e-4, ignore_index=C)
    with contextlib.redirect_stdout(io.StringIO()):
        m._train_epoch(1, model, tr, 1000)          # warm-up epoch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e in range(2, 4):
            m._train_epoch(e, model, tr, 1000)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"driver train loop: {2 * (N // 4) * 4 / dt:7.1f} images/s ({dt / (2 * (N // 4)) * 1e3:.2f} ms/step)")
    t0 = time.perf_counter()
    for _ in mk(ds, 4, True):
        pass
    print(f"dataloader alone : {(time.perf_counter() - t0) / (N // 4) * 1e3:.2f} ms/batch")
