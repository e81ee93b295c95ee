#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> one row per (kernel, launch shape): count, avg / min / max / total microseconds.

`rocprofv3 --stats` averages every launch of a kernel name: a leg that launches scan_topk_kernel over 1 M rows (warm-up, c2) and
over 100 M rows (c4), or K1 over Zipf and over uniform ids, gets ONE row whose average describes neither.  Here a
shape is (kernel name, grid size, workgroup size) and, inside that, a cluster of durations: the sorted durations are cut wherever
two neighbours differ by more than --gap (default 1.35 x) -- launches of one shape over different amounts of data.  Clusters of fewer
than --min-count launches are folded into an "other" row.  Usage: kernel_shapes.py <kernel_trace.csv> [--only smt::] > shapes.csv"""
import argparse
import collections
import csv
import sys

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--only", default="smt::", help="keep kernels whose name contains this")
ap.add_argument("--gap", type=float, default=1.35)
ap.add_argument("--min-count", type=int, default=2)
a = ap.parse_args()

groups = collections.defaultdict(list)
for r in csv.DictReader(open(a.trace)):
    name = r["Kernel_Name"]
    if a.only and a.only not in name:
        continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    short = name.replace("void smt::", "").replace("smt::", "").split("(")[0]
    grid = "x".join(v for v in (r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", "")) if v not in ("", "1")) or "1"
    wg = "x".join(v for v in (r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("Workgroup_Size_Y", ""), r.get("Workgroup_Size_Z", "")) if v not in ("", "1")) or "1"
    groups[(short, grid, wg)].append(dur)

w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid", "workgroup", "cluster", "launches", "avg_us", "min_us", "max_us", "total_us"])
rows = []
for (name, grid, wg), durs in groups.items():
    durs.sort()
    clusters, cur = [], [durs[0]]
    for d in durs[1:]:
        if d > cur[-1] * a.gap:
            clusters.append(cur)
            cur = [d]
        else:
            cur.append(d)
    clusters.append(cur)
    small = [d for c in clusters if len(c) < a.min_count for d in c]
    big = [c for c in clusters if len(c) >= a.min_count]
    for i, c in enumerate(big):
        rows.append((name, grid, wg, f"{i + 1}/{len(big)}", len(c), sum(c) / len(c), c[0], c[-1], sum(c)))
    if small:
        rows.append((name, grid, wg, "other", len(small), sum(small) / len(small), min(small), max(small), sum(small)))
rows.sort(key=lambda r: -r[8])
for r in rows:
    w.writerow([r[0], r[1], r[2], r[3], r[4]] + [f"{v:.2f}" for v in r[5:]])
