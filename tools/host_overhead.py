import os, sys, time, warnings, cProfile, pstats
from argparse import Namespace
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelpick_amd.utils.utils import get_model
from pixelpick_amd.trainer import FlatTrainer
from bench import synth_train_batch
warnings.simplefilter("ignore")
m = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=19, network_name="deeplab", weight_type="random")).cuda().train()
tr = FlatTrainer(m, ignore_index=19)
x, y = synth_train_batch(4, 19, 256, 512, 20, torch.device("cuda"), 1)
for _ in range(3): tr.train_step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): tr.train_step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/20:.2f} ms/step, total {1e3*(t2-t0)/20:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): tr.train_step(x, y)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(50)
