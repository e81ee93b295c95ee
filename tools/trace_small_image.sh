#!/usr/bin/env bash
# kernel timeline of ONE 1-query / 8-query call over the operand image (10 M rows): durations and the gaps between launches
root="$(pwd)"; out="$root/gpurun_out"; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace_small" -o t -- python "$root/tools/image_scan_probe.py" > "$out/trace_small.log" 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/trace_small/t_kernel_trace.csv")))
rows=[r for r in rows if r["Kernel_Name"].startswith(("smt::","void smt::"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last call of the 8-query run: take the final 14 smt kernels
last=rows[-14:]
t0=int(last[0]["Start_Timestamp"])
prev_end=None
for r in last:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%8.1f us  +%6.1f gap  %8.1f us  %s" % ((s-t0)/1e3, 0 if prev_end is None else (s-prev_end)/1e3, (e-s)/1e3, r["Kernel_Name"][:60]))
    prev_end=e
print("span %.1f us" % ((int(last[-1]["End_Timestamp"])-t0)/1e3))
PY
