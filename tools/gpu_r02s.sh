#!/usr/bin/env bash
# round 2, after the K3 rewrite (bf16 x 3 + row-register kernel): tests, bench line, K3 evidence
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"; out="$root/gpurun_out"; mkdir -p "$out"
python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -4
python __graft_entry__.py smoke 2>&1 | tail -2
timeout 1200 python bench.py > "$out/bench_r02s.json" 2> "$out/bench_r02s.err"; cut -c1-3000 "$out/bench_r02s.json"; tail -3 "$out/bench_r02s.err"
: > "$out/k3_modes.jsonl"
for mode in "gemm_bf16x3=1" "gemm_rowreg=0" "gemm_bf16x3=0"; do
  timeout 600 python tools/bench_small_batch.py --nq 8 32 64 128 256 1000 --variants 1 --reps 7 --tune $mode 2>&1 | grep -E "^\{" | sed "s/^{/{\"mode\": \"$mode\", /" >> "$out/k3_modes.jsonl"
done
cut -c1-200 "$out/k3_modes.jsonl"
timeout 200 tools/micro/row_load_patterns > "$out/row_load_patterns.jsonl"; cat "$out/row_load_patterns.jsonl"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_k3rr" -o k3 -- python "$root/tools/bench_small_batch.py" --nq 32 128 1000 --variants 1 --reps 5 > "$out/prof_k3rr.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_k3rr_fetch" -o k3 -- python "$root/tools/bench_small_batch.py" --nq 32 128 --variants 1 --reps 2 > "$out/pmc_k3rr_fetch.log" 2>&1
python "$root/tools/summarize_pmc.py" "$out/pmc_k3rr_fetch" "$out/k3rr_traffic.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/bench_small_batch.py --nq 32 128 --variants 1 --reps 2 (10 M rows: 10.24 GB algorithmic per corpus pass; the main level reads 15/16 of it)" | cut -c1-1200
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$out/pmc_k3rr_mfma" -o k3 -- python "$root/tools/bench_small_batch.py" --nq 1000 --variants 1 --reps 2 > "$out/pmc_k3rr_mfma.log" 2>&1
python "$root/tools/summarize_pmc.py" "$out/pmc_k3rr_mfma" "$out/k3rr_mfma.json" "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -- python tools/bench_small_batch.py --nq 1000 --variants 1 --reps 2" | cut -c1-1500
find "$out" -name "*kernel_stats.csv" | head
find "$out" -name "*.csv" -size +6M -delete
