#!/usr/bin/env bash
# round 5: kept range sets + batched calls without the overflow read-back: tests, then the workspace / c3 legs of the bench
out="$(pwd)/gpurun_out"; mkdir -p "$out"
timeout 1500 python -m pytest tests/test_gpu_filtered_batches.py tests/test_gpu_batched.py tests/test_gpu_nearties.py tests/test_gpu_scan.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -12
timeout 900 python bench.py --steps 50 --warmup 10 --no-embed --no-ingest --no-group-issue --no-c4 --no-cpu-baseline --no-ivfpq \
   --detail-out "$out/r05d_detail.json" > "$out/r05d_line.json" 2> "$out/r05d.err"
python - "$out/r05d_line.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in sorted(d):
    if k.startswith(("ws_", "c3_", "checks")): print(k, d[k])
PY
