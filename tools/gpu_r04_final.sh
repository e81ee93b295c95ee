#!/usr/bin/env bash
# Round 4 checkpoint / closing run: the whole GPU suite, smoke(), the default bench line + detail, per-leg rocprofv3 kernel stats
# (one file per leg, so that every `frac` of the line reproduces from a tracked file: VERDICT r3 weak 7), FETCH_SIZE of the legs.
# Usage: bash tools/gpu_r04_final.sh <tag>
tag="${1:-r04}"
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
step() { echo "=== $1 ($(date +%T))"; }
step "pytest -m gpu"
timeout 1500 python -m pytest tests -q -m gpu -x > "$out/${tag}_gpu_suite.log" 2>&1; echo "suite rc=$?"; tail -3 "$out/${tag}_gpu_suite.log"
step "smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
step "bench (default)"
timeout 900 python bench.py --detail-out "$out/${tag}_bench_detail.json" > "$out/${tag}_bench_line.json" 2> "$out/${tag}_bench.err"; echo "bench rc=$? line bytes=$(tail -1 "$out/${tag}_bench_line.json" | wc -c)"
tail -1 "$out/${tag}_bench_line.json"; grep -v "bench detail" "$out/${tag}_bench.err" | tail -3
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench flags...
  name="$1"; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}_$name" -o bench -- python "$root/bench.py" --no-cpu-baseline --detail-out "$out/${tag}_bench_detail_prof_$name.json" "$@" > "$out/prof_${tag}_$name.log" 2>&1
  cp "$out/prof_${tag}_$name/bench_kernel_stats.csv" "$out/${tag}_bench_${name}_kernel_stats.csv" 2>/dev/null
  head -4 "$out/${tag}_bench_${name}_kernel_stats.csv" | cut -c1-160
}
step "kernel stats per leg"
stats c2 --steps 1000 --warmup 100 --no-c4 --no-secondary --no-embed --no-ivfpq --no-workspace --no-ingest --no-group-issue
stats c4 --steps 20 --warmup 5 --no-secondary --no-embed --no-ivfpq --no-workspace --no-ingest --no-group-issue
stats c3 --steps 20 --warmup 5 --no-c4 --no-embed --no-ivfpq --no-workspace --no-ingest --no-group-issue
stats workspace --steps 20 --warmup 5 --no-c4 --no-secondary --no-embed --no-ivfpq --no-ingest --no-group-issue
stats embed --steps 20 --warmup 5 --no-c4 --no-secondary --no-ivfpq --no-workspace --no-ingest --no-group-issue
stats ivfpq --steps 20 --warmup 5 --no-c4 --no-secondary --no-embed --no-workspace --no-ingest --no-group-issue
step "FETCH_SIZE"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch_$tag" -o bench -- python "$root/bench.py" --steps 20 --warmup 3 --settle-steps 8 --no-cpu-baseline --no-ivfpq --no-ingest --no-group-issue --c4-steps 3 --detail-out "$out/${tag}_bench_detail_pmc.json" > "$out/pmc_fetch_$tag.log" 2>&1
python "$root/tools/summarize_pmc.py" "$out/pmc_fetch_$tag" "$out/${tag}_pmc_fetch.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 20 --warmup 3 --settle-steps 8 --no-cpu-baseline --no-ivfpq --no-ingest --no-group-issue --c4-steps 3" > /dev/null 2>&1
python - "$out/${tag}_pmc_fetch.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["kernels"].items():
    if "FETCH_SIZE_avg" in v: print(k[:70], v["dispatches"], {a: round(v[a] / 1e9, 4) for a in ("hbm_bytes_avg_corrected_x2", "hbm_bytes_max_corrected_x2")}, "min GB", round(2048 * v["FETCH_SIZE_min"] / 1e9, 4))
PY
step "K2 / K3 crossover sweep"
cd "$root" && timeout 300 python tools/sweep_k2_k3_small.py 2>&1 | tail -40
find "$out" -name "*.csv" -size +8M -delete
step "done"
