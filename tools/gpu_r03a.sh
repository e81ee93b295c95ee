#!/usr/bin/env bash
# Round-3 measurement session A (run on the GPU box from the repo root): MFMA rounding probe, c5 at its real size,
# rocprofv3 kernel stats + FETCH_SIZE counters of the bench legs.  Outputs under gpurun_out/ (copied to profiles/ afterwards).
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
tools/micro/mfma_rounding > "$out/r03_mfma_rounding.json" 2> "$out/r03_mfma_rounding.err"; tail -c 600 "$out/r03_mfma_rounding.json"
timeout 900 python bench.py --steps 50 --warmup 10 --settle-steps 16 --no-c4 --no-secondary --no-embed --no-cpu-baseline --c5-rows 100000000 \
    > "$out/r03_bench_c5_100M.json" 2> "$out/r03_bench_c5_100M.err"; tail -2 "$out/r03_bench_c5_100M.err"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench_c5_100M.json"))["ivfpq"]
print({k: d.get(k) for k in ("build_s", "build_ms", "index_bytes", "recall_at_k_vs_exact", "queries_per_s")}, d.get("global_pq_m32", {}).get("recall_at_k_vs_exact"), d.get("error"))
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_r03" -o bench -- python "$root/bench.py" --steps 1000 --warmup 100 --no-cpu-baseline --c4-steps 10 > "$out/prof_r03.log" 2>&1
tail -2 "$out/prof_r03.log"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch_r03" -o bench -- python "$root/bench.py" --steps 20 --warmup 3 --settle-steps 8 --no-cpu-baseline --c4-steps 3 > "$out/pmc_fetch_r03.log" 2>&1
tail -2 "$out/pmc_fetch_r03.log"
cd "$root"
python tools/summarize_pmc.py "$out/pmc_fetch_r03" "$out/r03_pmc_fetch.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 20 --warmup 3 --settle-steps 8 --no-cpu-baseline --c4-steps 3" > /dev/null
cp "$(find "$out/prof_r03" -name '*kernel_stats.csv' | head -1)" "$out/r03_bench_kernel_stats.csv" 2>/dev/null
find "$out" -name "*.csv" -size +8M -delete
find "$out" -name "*.db" -delete
ls -la "$out" | tail -20
