#!/usr/bin/env bash
# Kernel timeline of the LAST 1000-query IVF search of tools/ivf_call.py at ROWS rows (the 30 M-row cliff).
# Usage: ROWS=30000000 bash tools/trace_ivf.sh <tag>
tag="$1"; root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"; cd /tmp && export TMPDIR=/tmp
rm -rf "$out/trace_$tag"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$out/trace_$tag" -o t -- python "$root/tools/ivf_call.py" > "$out/trace_$tag.log" 2>&1
tail -1 "$out/trace_$tag.log"
python - "$out/trace_$tag" "$tag" <<'PY'
import csv, glob, json, sys
d, tag = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "smt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
heads = [i for i, n in enumerate(names) if "ivf_score_kernel" in n]
last = rows[heads[-1]:]
t0 = int(last[0]["Start_Timestamp"]); prev = None; tl = []
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tl.append({"at_us": round((s - t0) / 1e3, 1), "gap_us": 0.0 if prev is None else round((s - prev) / 1e3, 1), "dur_us": round((e - s) / 1e3, 1),
               "kernel": r["Kernel_Name"].replace("void smt::", "").replace("smt::", "")[:60], "grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size")})
    prev = e
span = (int(last[-1]["End_Timestamp"]) - t0) / 1e3
json.dump({"span_us": round(span, 1), "kernels": tl}, open(d + "/../timeline_" + tag + ".json", "w"), indent=1)
for k in tl: print("%8.1f  +%6.1f  %8.1f  %s  grid %s" % (k["at_us"], k["gap_us"], k["dur_us"], k["kernel"], k["grid"]))
print("span %.1f us" % span)
PY
find "$out/trace_$tag" -name "*.csv" -size +4M -delete
