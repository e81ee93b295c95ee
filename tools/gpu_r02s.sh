#!/usr/bin/env bash
# round 2 evidence run: tests, smoke, the bench line, K3 batch times per nomination mode, rocprofv3 stats / counters.
# Every step runs under its own timeout (a stuck step must not eat the GPU budget).
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"; out="$root/gpurun_out"; mkdir -p "$out"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > "$out/pytest_gpu.log" 2>&1; grep -E "passed|failed" "$out/pytest_gpu.log" | tail -2
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 1200 python bench.py > "$out/bench_r02s.json" 2> "$out/bench_r02s.err"; cut -c1-1500 "$out/bench_r02s.json"; tail -2 "$out/bench_r02s.err"
: > "$out/k3_modes.jsonl"
for mode in "gemm_nominate=0" "gemm_nominate=1" "gemm_nominate=2" "gemm_rowreg=0" "gemm_bf16x3=0"; do
  timeout 400 python tools/bench_small_batch.py --nq 8 32 64 128 256 1000 --variants 1 --reps 7 --tune $mode 2>&1 | grep -E "^\{" | sed "s/^{/{\"mode\": \"$mode\", /" >> "$out/k3_modes.jsonl"
done
cut -c1-125 "$out/k3_modes.jsonl"
timeout 200 tools/micro/row_load_patterns > "$out/row_load_patterns.jsonl"
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-ivfpq --no-c4"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_c2" -o bench -- python "$root/bench.py" --steps 1000 --warmup 100 $B > "$out/prof_c2.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_c2" -o bench -- python "$root/bench.py" --steps 20 --warmup 3 --settle-steps 8 $B > "$out/pmc_c2.log" 2>&1
timeout 100 python "$root/tools/summarize_pmc.py" "$out/pmc_c2" "$out/traffic_c2.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 20 --warmup 3 --settle-steps 8 $B" | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_k3rr" -o k3 -- python "$root/tools/bench_small_batch.py" --nq 32 128 1000 --variants 1 --reps 5 > "$out/prof_k3rr.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_k3rr_fetch" -o k3 -- python "$root/tools/bench_small_batch.py" --nq 32 128 --variants 1 --reps 2 > "$out/pmc_k3rr_fetch.log" 2>&1
timeout 100 python "$root/tools/summarize_pmc.py" "$out/pmc_k3rr_fetch" "$out/k3rr_traffic.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/bench_small_batch.py --nq 32 128 --variants 1 --reps 2 (10 M rows: 10.24 GB algorithmic per corpus pass; the main level reads 15/16 of it)" | cut -c1-200
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$out/pmc_k3rr_mfma" -o k3 -- python "$root/tools/bench_small_batch.py" --nq 1000 --variants 1 --reps 2 > "$out/pmc_k3rr_mfma.log" 2>&1
timeout 100 python "$root/tools/summarize_pmc.py" "$out/pmc_k3rr_mfma" "$out/k3rr_mfma.json" "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -- python tools/bench_small_batch.py --nq 1000 --variants 1 --reps 2" | cut -c1-200
find "$out" -name "*kernel_stats.csv" | head
find "$out" -name "*.csv" -size +6M -delete
