#!/usr/bin/env python3
"""Timeline of one train step from a rocprofv3 --kernel-trace CSV: per HIP queue busy / idle time, and the kernels of the
last complete step in launch order with their start offsets, so that gaps and cross-stream overlap can be read off.

    python tools/timeline.py <kernel_trace.csv> [--list] [--step-marker nchw_to_nhwc]
"""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "").replace("pp::", "")
    return n.split("(")[0][:44]


def main():
    path = sys.argv[1]
    marker = "nchw_to_nhwc"
    if "--step-marker" in sys.argv:
        marker = sys.argv[sys.argv.index("--step-marker") + 1]
    if path.endswith(".db"):                      # rocprofv3's default rocpd output
        import sqlite3
        cur = sqlite3.connect(path).cursor()
        rows = [dict(zip(("Kernel_Name", "Queue_Id", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y",
                          "Grid_Size_Z", "Workgroup_Size_X"), r)) for r in
                cur.execute("select name, queue_id, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels")]
    else:
        rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    if len(starts) < 3:
        sys.exit("not enough steps in the trace")
    a, b = starts[-2], starts[-1]
    step = rows[a:b]
    t0, t1 = step[0]["s"], rows[b]["s"]
    print(f"step wall {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels")
    byq = defaultdict(list)
    for r in step:
        byq[r["Queue_Id"]].append(r)
    for q, rs in sorted(byq.items()):
        busy = sum(r["e"] - r["s"] for r in rs)
        gaps = [(rs[i + 1]["s"] - rs[i]["e"]) for i in range(len(rs) - 1)]
        small = sum(g for g in gaps if 0 < g < 20000)
        print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e3:.1f} us, span {(rs[-1]['e'] - rs[0]['s']) / 1e3:.1f} us, "
              f"gaps<20us total {small / 1e3:.1f} us (n={sum(1 for g in gaps if 0 < g < 20000)}), "
              f"median gap {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.2f} us")
    # union busy (any queue) -> GPU idle inside the step
    ev = sorted((r["s"], r["e"]) for r in step)
    cur_s, cur_e, union = ev[0][0], ev[0][1], 0
    for s, e in ev[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print(f"GPU busy (any queue) {union / 1e3:.1f} us, idle {(t1 - t0 - union) / 1e3:.1f} us")
    agg = defaultdict(lambda: [0, 0])
    for r in step:
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += r["e"] - r["s"]
    print("-- per kernel (this step)")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"  {k:46s} n={n:3d} {t / 1e3:8.1f} us")
    if "--by-queue" in sys.argv:
        for q, rs in sorted(byq.items()):
            aq = defaultdict(lambda: [0, 0])
            for r in rs:
                k = short(r["Kernel_Name"])
                aq[k][0] += 1
                aq[k][1] += r["e"] - r["s"]
            print(f"-- queue {q}: every kernel")
            for k, (n, t) in sorted(aq.items(), key=lambda kv: -kv[1][1]):
                print(f"  {k:46s} n={n:3d} {t / 1e3:8.1f} us  ({t / n / 1e3:6.1f} each)")
    if "--list" in sys.argv:
        print("-- launch order: start offset us | dur us | queue | grid | kernel")
        for r in step:
            g = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            print(f"  {(r['s'] - t0) / 1e3:8.1f} {(r['e'] - r['s']) / 1e3:7.1f}  q{r['Queue_Id']} {g:6d}  {short(r['Kernel_Name'])}")


if __name__ == "__main__":
    main()
