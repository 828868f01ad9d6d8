#!/usr/bin/env python3
"""bench.py — PixelPick hot path on MI355X: one JSON line per run (driver contract).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the acquisition hot path (softmax -> uncertainty -> exclusion -> per-image
top-k, query.py:190-204,57-61) over one batch of synthetic logits already resident in HBM.
Workload = BASELINE.json configs[1]: Cityscapes-quarter 256x512, C=19, entropy, top-k=20, batched
B=256 images per launch per GPU (2.58 GB of logits >> the 256 MiB Infinity Cache), NCHW fp32 as the
reference model emits.  Images shard over ranks with no data-path collective ("weak" scaling).

Extra objects on the line:
  roofline     HBM roofline of the dominant kernel (acq_kernel): algorithmic bytes per launch
               = pixels*(4*C+1) / its mean launch duration measured with HIP events
               recorded around that kernel on its stream (pp_debug_set_kernel_events).
  cpu_baseline the torch-CPU port of the reference path (oracle/acq.py) timed on this box's host
               cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


class HipEvents:
    """n start/stop hipEvent pairs created through the HIP runtime PyTorch already loaded."""

    def __init__(self, n):
        self.hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.n = n
        self.starts = (ctypes.c_void_p * n)()
        self.stops = (ctypes.c_void_p * n)()
        for arr in (self.starts, self.stops):
            for i in range(n):
                ev = ctypes.c_void_p()
                rc = self.hip.hipEventCreate(ctypes.byref(ev))
                assert rc == 0, f"hipEventCreate -> {rc}"
                arr[i] = ev

    def elapsed_ms(self):
        out = []
        for i in range(self.n):
            ms = ctypes.c_float()
            rc = self.hip.hipEventElapsedTime(ctypes.byref(ms), self.starts[i], self.stops[i])
            assert rc == 0, f"hipEventElapsedTime -> {rc}"
            out.append(ms.value)
        return out

    def destroy(self):
        for arr in (self.starts, self.stops):
            for i in range(self.n):
                self.hip.hipEventDestroy(arr[i])


def cpu_baseline(C, H, W, k, strategy, budget_s=12.0):
    """Reference path on host cores: per-image loop like query.py:159 (B=1), torch CPU ops."""
    from oracle import acq as orc
    threads = torch.get_num_threads()
    gen = torch.Generator().manual_seed(0)
    n_img = 4
    logits = [torch.randn(1, C, H, W, generator=gen) * 3 for _ in range(n_img)]
    excl = [torch.rand(H, W, generator=gen) < 0.05 for _ in range(n_img)]
    orc.torch_port_acquire(logits[0], excl[0], strategy, k)  # warm-up
    t0 = time.perf_counter()
    done = 0
    while True:
        orc.torch_port_acquire(logits[done % n_img], excl[done % n_img], strategy, k)
        done += 1
        el = time.perf_counter() - t0
        if el > budget_s or done >= 2000:
            break
    return {"value": round(done * H * W / el / 1e6, 3), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "sample": f"{done} images {H}x{W}x{C} one at a time (query.py:159 loop), torch CPU ops, "
                      f"{threads} threads, {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per launch per GPU")
    ap.add_argument("--classes", type=int, default=19)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--strategy", default="entropy")
    ap.add_argument("--layout", default="nchw", choices=["nchw", "nhwc"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tune-occ", type=int, default=0)
    ap.add_argument("--tune-ppt", type=int, default=0)
    ap.add_argument("--reduce-mode", type=int, default=0)
    ap.add_argument("--exact-formula", type=int, default=0)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 or world > 1:
        import torch.distributed as dist
        assert world == a.gpus, f"WORLD_SIZE={world} but --gpus {a.gpus}: launch with torch.distributed.run"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    from pixelpick_amd import _lib
    from pixelpick_amd import acquisition as acq
    L = _lib.lib()
    L.pp_debug_set_acq_tuning(a.tune_occ, a.tune_ppt)
    L.pp_debug_set_reduce_mode(a.reduce_mode)
    L.pp_debug_set_exact_formula(a.exact_formula)

    B, C, H, W, k = a.batch, a.classes, a.height, a.width, a.k
    gen = torch.Generator(device=dev).manual_seed(rank)
    logits = torch.randn((B, C, H, W), device=dev, generator=gen) * 3          # resident in HBM before timing
    if a.layout == "nhwc":
        logits = logits.contiguous(memory_format=torch.channels_last)
    excl = (torch.rand((B, H, W), device=dev, generator=gen) < 0.05).to(torch.uint8)

    # pre-allocate everything the step touches: the timed region is kernels only
    idx = torch.empty((B, k), dtype=torch.int32, device=dev)
    val = torch.empty((B, k), dtype=torch.float32, device=dev)
    ws = torch.empty(max(L.pp_acq_workspace_bytes(B, C, H, W, k), 256), dtype=torch.uint8, device=dev)
    sB, sC, sH, sW = logits.stride()
    stream = torch.cuda.current_stream(dev).cuda_stream
    sid = acq.STRATEGY_ID[a.strategy]

    def step():
        rc = L.pp_acq_score_topk(logits.data_ptr(), B, C, H, W, sB, sC, sH, sW, excl.data_ptr(), sid, k,
                                 idx.data_ptr(), val.data_ptr(), None, ws.data_ptr(), ws.numel(), stream)
        _lib.check(rc, "pp_acq_score_topk")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    ev = HipEvents(a.steps)
    barrier()
    L.pp_debug_set_kernel_events(ev.starts, ev.stops, a.steps)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    el = time.perf_counter() - t0
    L.pp_debug_set_kernel_events(None, None, 0)
    kern_ms = ev.elapsed_ms()
    ev.destroy()

    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    if rank == 0:
        pixels_per_step = world * B * H * W
        ms_per_step = el / a.steps * 1e3
        value = pixels_per_step / (el / a.steps) / 1e6
        alg_bytes = B * H * W * (4 * C + 1)                 # per launch of acq_kernel on ONE GPU
        kavg_ms = sum(kern_ms) / len(kern_ms)
        achieved = alg_bytes / (kavg_ms * 1e-3) / 1e9
        line = {
            "metric": "acquisition_throughput", "value": round(value, 1), "unit": "Mpixels/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: Cityscapes {H}x{W}, C={C}, {a.strategy} acquisition "
                                   f"top-k={k}, logits of DeepLabv3+-MNv2 shape, {a.layout.upper()} fp32",
                       "images_per_launch_per_gpu": B, "k": k, "strategy": a.strategy, "layout": a.layout,
                       "sharding": f"images over {world} rank(s), no collective"},
            "roofline": {"bound": "hbm", "kernel": "acq_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_avg": round(kavg_ms, 4),
                         "kernel_ms_min": round(min(kern_ms), 4)},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(C, H, W, k, a.strategy)
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
