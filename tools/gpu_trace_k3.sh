#!/usr/bin/env bash
# per-dispatch durations of one K3 batch (kernel trace only)
tag="${1:-r01}"; shift
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/trace_c3_$tag" -o c3 -- python "$root/tools/bench_batched.py" --reps 1 --check 1 "$@" > "$out/trace_c3_$tag.log" 2>&1
python - "$out/trace_c3_$tag" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("smt::")]
for r in rows[-16:]:
    print(r["Kernel_Name"][:40], "grid", r["Grid_Size_X"], "us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
