#!/usr/bin/env bash
# HBM traffic of the bench legs' dominant kernels on the round-6 binary: ONE rocprofv3 --pmc FETCH_SIZE pass (--kernel-trace only beside
# it) over a short bench run, summarised by tools/summarize_pmc.py; tools/collect_traffic.py then folds it into profiles/traffic.json.
# Usage (GPU box, repo root): bash tools/gpu_r06_fetch.sh
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cmd="--steps 20 --warmup 3 --settle-steps 8 --no-cpu-baseline --no-ivfpq --no-ingest --no-group-issue --no-small-calls --c4-steps 3"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch_r06" -o bench -- python "$root/bench.py" $cmd --detail-out "$out/r06_bench_detail_pmc.json" > "$out/pmc_fetch_r06.log" 2>&1
python "$root/tools/summarize_pmc.py" "$out/pmc_fetch_r06" "$out/r06_pmc_fetch.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py $cmd" > /dev/null 2>&1
find "$out" -name "*.csv" -size +8M -delete
python - <<'PY'
import json
d = json.load(open("/root/repo/gpurun_out/r06_pmc_fetch.json"))
for k, v in d["kernels"].items():
    if "FETCH_SIZE_avg" in v:
        print(k[:70], v["dispatches"], {a: round(b, 1) for a, b in v.items() if a.startswith("FETCH_SIZE") or a.startswith("hbm_bytes")})
PY
