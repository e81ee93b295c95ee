#!/usr/bin/env bash
# Kernel timeline of the LAST call of tools/k3_call.py "$@": start offset, gap to the previous kernel, duration, name.
# Usage (GPU box, repo root): bash tools/trace_call.sh <tag> [k3_call.py arguments]
tag="$1"; shift
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"; cd /tmp && export TMPDIR=/tmp
rm -rf "$out/trace_$tag"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace_$tag" -o t -- python "$root/tools/k3_call.py" "$@" > "$out/trace_$tag.log" 2>&1
tail -1 "$out/trace_$tag.log"
python - "$out/trace_$tag" "$tag" "$*" <<'PY'
import csv, glob, json, sys
d, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "smt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the calls repeat the same kernel sequence: find the period from the end (first kernel name of a call = the last call's first kernel)
names = [r["Kernel_Name"] for r in rows]
heads = [i for i, n in enumerate(names) if n.split("(")[0].endswith(("split_queries_f16x1_kernel", "split_queries_f16_kernel", "split_queries_kernel", "build_chunk_table_kernel", "build_tile_table_kernel"))]
start = heads[-1] if heads else max(0, len(rows) - 16)
if heads and "table_kernel" in names[start] and len(heads) > 1 and heads[-2] == start - 1: start -= 1
last = rows[start:]
t0 = int(last[0]["Start_Timestamp"]); prev = None; tl = []
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tl.append({"at_us": round((s - t0) / 1e3, 1), "gap_us": 0.0 if prev is None else round((s - prev) / 1e3, 1), "dur_us": round((e - s) / 1e3, 1),
               "kernel": r["Kernel_Name"].replace("void smt::", "").replace("smt::", "")[:70], "grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size")})
    prev = e
span = (int(last[-1]["End_Timestamp"]) - t0) / 1e3
res = {"args": args, "span_us": round(span, 1), "busy_us": round(sum(k["dur_us"] for k in tl), 1), "gaps_us": round(sum(k["gap_us"] for k in tl), 1), "kernels": tl}
json.dump(res, open(d + "/../timeline_" + tag + ".json", "w"), indent=1)
for k in tl: print("%8.1f  +%6.1f  %8.1f  %s  grid %s" % (k["at_us"], k["gap_us"], k["dur_us"], k["kernel"], k["grid"]))
print("span %.1f us, busy %.1f, gaps %.1f" % (span, res["busy_us"], res["gaps_us"]))
PY
find "$out/trace_$tag" -name "*.csv" -size +4M -delete
