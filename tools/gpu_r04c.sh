#!/usr/bin/env bash
# Round 4, session C: the bootstrap level plan of K3 -- parity, timelines with and without it on the same box, workspace leg.
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
step() { echo "=== $1 ($(date +%T))"; }
step "pytest K3 paths"
timeout 1200 python -m pytest tests/test_gpu_batched.py tests/test_gpu_filtered_batches.py tests/test_gpu_image.py tests/test_gpu_nearties.py tests/test_gpu_defaults.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "^E |passed|failed|Error" | head -20
step "timelines (bootstrap plan | appended-levels plan)"
for cfg in "1000_img --nq 1000" "1000_f32 --nq 1000 --no-image" "256_img --nq 256 --reps 5" "8_img --nq 8 --reps 10" "1_img --nq 1 --reps 10" "256_subset_img --nq 256 --reps 5 --subset"; do
  tag="${cfg%% *}"; args="${cfg#* }"
  bash tools/trace_call.sh "${tag}_boot" $args | tail -12
  bash tools/trace_call.sh "${tag}_old" $args --tune gemm_bootstrap=0 | tail -3
done
step "bench workspace + c3 legs"
timeout 400 python bench.py --steps 200 --warmup 50 --no-c4 --no-embed --no-ivfpq --no-ingest --no-cpu-baseline --detail-out "$out/bench_detail_r04c.json" 2> "$out/bench_r04c.err" | tail -c 3500; grep -v "bench detail" "$out/bench_r04c.err" | tail -3
step "done"
