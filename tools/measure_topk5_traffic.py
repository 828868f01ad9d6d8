#!/usr/bin/env python3
"""HBM traffic of the scorer launch in the reference's default top-5 % mode (args.py:25), PMC counters, the way tools/measure_acq_traffic.py
measures the k = 20 scorer (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes with --kernel-trace only; FETCH doubled for gfx950 as
MI355X_MICROARCH.md prescribes): the list select's emitting scorer (default) against the map-writing scorer (RMODE=2048).  GPU box:

    python tools/measure_topk5_traffic.py          # prints a table; nothing is written into profiles/ (copy what you want judged)
"""
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "topk5_pmc")
B, C, H, W = 256, 19, 256, 512


def one_pass(counter, rmode):
    d = os.path.join(OUT, f"{counter}_{rmode}")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    env = dict(os.environ, RMODE=str(rmode), MASK="1", TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                    sys.executable, os.path.join(ROOT, "tools", "topk5_bench.py")], cwd="/tmp", env=env, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            n = r["Kernel_Name"].split("(")[0]
            if "pp::" not in n:
                continue
            per.setdefault(n, []).append(float(r["Counter_Value"]))
    # one row per dispatch; the first launch of a kernel also pages its buffers in
    return {n: sum(v[1:]) / len(v[1:]) for n, v in per.items() if len(v) > 1}


def main():
    alg = B * H * W * (4 * C + 1)
    print(f"B={B} {H}x{W}x{C}, k = 5 %: algorithmic bytes of the scorer launch {alg / 1e6:.1f} MB (4 C + 1 per pixel); score map {B * H * W * 4 / 1e6:.1f} MB")
    for rmode, name in ((0, "list select (default)"), (2048, "map path (pp_debug_set_reduce_mode bit 11)")):
        fetch = one_pass("FETCH_SIZE", rmode)
        write = one_pass("WRITE_SIZE", rmode)
        print(f"-- {name}")
        for n in sorted(set(fetch) | set(write)):
            f = fetch.get(n, 0.0) * 1024 * 2            # KB per launch, doubled (gfx950 correction)
            w = write.get(n, 0.0) * 1024
            print(f"   {n[:110]:110s} fetch {f / 1e6:9.1f} MB   write {w / 1e6:8.1f} MB per launch")


if __name__ == "__main__":
    main()
