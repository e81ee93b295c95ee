#!/usr/bin/env bash
# kernel timeline of bench.py's N>1 path forced onto one rank: per-kernel average duration and the scan-to-scan period
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
SEMTOOLS_BENCH_FORCE_EXCHANGE=1 rocprofv3 --kernel-trace --output-format csv -d "$out/trace_xchg" -o t -- python "$root/bench.py" --steps 400 --warmup 50 --settle-steps 64 --no-secondary --no-ivfpq --no-cpu-baseline --no-c4 > "$out/trace_xchg.log" 2>&1
grep '"metric"' "$out/trace_xchg.log" | cut -c1-400
python - "$out/trace_xchg" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
scans = []
for r in rows:
    n = r["Kernel_Name"][:70]
    agg.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if "scan_topk_kernel" in n:
        scans.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in agg.items():
    v2 = v[len(v)//2:]
    print(f"{k:70s} n={len(v):5d} avg_us(last half)={sum(v2)/len(v2):9.1f}")
s2 = scans[len(scans)//2:]
per = [(b[0] - a[0]) / 1e3 for a, b in zip(s2, s2[1:])]
gap = [(b[0] - a[1]) / 1e3 for a, b in zip(s2, s2[1:])]
print("scan-to-scan period us: avg %.1f  min %.1f  max %.1f ; idle gap between scans avg %.1f" % (sum(per)/len(per), min(per), max(per), sum(gap)/len(gap)))
PY
find "$out/trace_xchg" -name "*.csv" -size +4M -delete
