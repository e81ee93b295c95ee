#!/usr/bin/env bash
# Round 4, session G: slot-folded bootstrap + ballot tau (parity, timelines), ingest after the parallel split.
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
step() { echo "=== $1 ($(date +%T))"; }
step "pytest K3 + host paths"
timeout 1500 python -m pytest tests/test_gpu_batched.py tests/test_gpu_filtered_batches.py tests/test_gpu_image.py tests/test_gpu_nearties.py tests/test_gpu_defaults.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_host.py -x -q 2>&1 | grep -E "^E |passed|failed|Error" | head -20
step "timelines"
for cfg in "1000_img --nq 1000" "256_img --nq 256 --reps 5" "32_img --nq 32 --reps 10" "1_img --nq 1 --reps 10" "256_subset_img --nq 256 --reps 5 --subset"; do
  tag="${cfg%% *}"; args="${cfg#* }"
  bash tools/trace_call.sh "${tag}_boot" $args | tail -11
  bash tools/trace_call.sh "${tag}_old" $args --tune gemm_bootstrap=0 | tail -1
done
step "bench c3 + workspace + ingest"
timeout 400 python bench.py --steps 200 --warmup 50 --no-c4 --no-embed --no-ivfpq --no-cpu-baseline --detail-out "$out/bench_detail_r04g.json" 2> "$out/bench_r04g.err" | tail -c 2600; grep -v "bench detail" "$out/bench_r04g.err" | tail -2
step "done"
