"""Aggregate a rocprofv3 `*_counter_collection.csv` (and, if present, the `*_kernel_trace.csv` / `*kernel_stats.csv`
next to it) into a small JSON: per smt:: kernel the number of dispatches, the average of every collected counter and
the average duration.  FETCH_SIZE is reported raw (KB) and corrected (x2 on gfx950: the counter tallies 128-B requests
at 64 B, MI355X_MICROARCH.md section HBM)."""
import collections
import csv
import glob
import json
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    res = dict(note=note, kernels={})
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "smt::" not in n:
                continue
            agg[n[:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[n[:100]]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in agg.items():
            e = res["kernels"].setdefault(k, {})
            e["dispatches"] = len(v["_dur_us"])
            for c, vals in v.items():
                if c == "_dur_us":
                    e["avg_us_under_pmc"] = sum(vals) / len(vals)
                    e["max_us_under_pmc"] = max(vals)
                else:
                    e[c + "_avg"] = sum(vals) / len(vals)
                    e[c + "_max"] = max(vals)
                    e[c + "_min"] = min(vals)
                    if c == "FETCH_SIZE":
                        e["hbm_bytes_avg_corrected_x2"] = 2048.0 * sum(vals) / len(vals)
                        e["hbm_bytes_max_corrected_x2"] = 2048.0 * max(vals)
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "smt::" in r.get("Name", ""):
                e = res["kernels"].setdefault(r["Name"][:100], {})
                e.update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), total_ns=float(r["TotalDurationNs"]),
                         pct=float(r["Percentage"]))
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:3000])


if __name__ == "__main__":
    main()
