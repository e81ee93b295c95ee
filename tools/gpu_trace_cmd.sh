#!/usr/bin/env bash
# rocprofv3 kernel trace of an arbitrary python command; prints avg duration per smt kernel
tag="$1"; shift
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/trace_$tag" -o t -- python "$@" > "$out/trace_$tag.log" 2>&1
grep "max_distance\|{" "$out/trace_$tag.log" | grep -v "rocprofv3" | tail -8
python - "$out/trace_$tag" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "smt::" not in n and "rocprim" not in n: continue
    agg.setdefault(n[:90], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v2 = v[len(v)//2:]
    print(f"{k:90s} n={len(v):5d} avg_us(last half)={sum(v2)/len(v2):9.1f}")
PY
find "$out/trace_$tag" -name "*.csv" -size +4M -delete
