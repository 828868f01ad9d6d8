#!/usr/bin/env python3
"""HBM traffic of the acquisition kernel from the PMC counters, the way MI355X_MICROARCH.md (HBM section) prescribes:
FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (--kernel-trace only), FETCH_SIZE doubled (gfx950 tallies a wide
coalesced read at half its bytes).  Runs on the GPU box:

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && python tools/measure_acq_traffic.py

and writes profiles/acq_traffic.json, keyed by the kernel symbol and the SHA-256 of csrc/acq.hip; bench.py quotes the figure
in `roofline.traffic` only while that hash still matches the source it runs (a changed kernel reports null, not a stale number).
"""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT_DIR = os.path.join(ROOT, "gpurun_out", "acq_pmc")
KERNEL = "acq_kernel"                  # pp::acq_kernel<19, ...>: the planar (NCHW) scorer bench.py's default line times


def acq_source_hash():
    h = hashlib.sha256()
    for f in ("acq.hip", "pp_common.h"):
        h.update(open(os.path.join(ROOT, "pixelpick_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def one_pass(counter):
    d = os.path.join(OUT_DIR, counter)
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "acq", "--steps", "6", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    vals, name = [], None
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL + "<" in r["Kernel_Name"] and "lowres" not in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
                name = r["Kernel_Name"]
    if len(vals) < 3:
        raise SystemExit(f"{counter}: only {len(vals)} {KERNEL} launches found under {d}")
    vals = vals[1:]                      # the first launch also pages the logits in
    return sum(vals) / len(vals), len(vals), name


def main():
    fetch_kb, n_f, name = one_pass("FETCH_SIZE")
    write_kb, n_w, _ = one_pass("WRITE_SIZE")
    B, C, H, W = 256, 19, 256, 512
    traffic = 2.0 * fetch_kb * 1024 + write_kb * 1024
    alg = B * H * W * (4 * C + 1)
    rec = {"kernel": name, "config": {"B": B, "C": C, "H": H, "W": W, "k": 20, "strategy": "entropy", "layout": "nchw"},
           "FETCH_SIZE_KB_per_launch": round(fetch_kb, 2), "WRITE_SIZE_KB_per_launch": round(write_kb, 2),
           "launches_averaged": [n_f, n_w], "fetch_correction": "x2 (gfx950: wide coalesced reads are tallied at half their bytes)",
           "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "ratio": round(traffic / alg, 4),
           "acq_source_sha256": acq_source_hash(), "measured_unix": int(time.time()),
           "command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python bench.py --mode acq "
                      "--steps 6 --warmup 1 --no-cpu-baseline --no-other-configs (two separate passes)"}
    with open(os.path.join(ROOT, "profiles", "acq_traffic.json"), "w") as f:
        json.dump(rec, f, indent=1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "acq_traffic.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
