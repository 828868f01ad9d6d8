#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel call-duration histogram for the LAST `steps` train steps.
    python tools/trace_summary.py <kernel_trace.csv> <name substring> [calls per step]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
per = int(sys.argv[3]) if len(sys.argv) > 3 else 60
sel = [r for r in rows if pat in r["Kernel_Name"]]
sel.sort(key=lambda r: int(r["Start_Timestamp"]))
last = sel[-per:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
print(f"{pat}: {len(sel)} calls; last {per}: sum {sum(d):.1f} us")
print(" ".join(f"{v:.0f}" for v in d))
