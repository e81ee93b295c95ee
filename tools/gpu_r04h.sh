#!/usr/bin/env bash
# Round 4, session H: timelines of the final K3 plan (wave-per-two-queries tau kernel) on one box, then the closing run.
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
echo "=== timelines ($(date +%T))"
for cfg in "1000_img --nq 1000" "1000_f32 --nq 1000 --no-image" "256_img --nq 256 --reps 5" "32_img --nq 32 --reps 10" "1_img --nq 1 --reps 10" "256_subset_img --nq 256 --reps 5 --subset" "1000_1M_img --nq 1000 --rows 1000000 --reps 5"; do
  tag="${cfg%% *}"; args="${cfg#* }"
  bash tools/trace_call.sh "${tag}_boot" $args | tail -11
  bash tools/trace_call.sh "${tag}_old" $args --tune gemm_bootstrap=0 | tail -1
done
bash tools/gpu_r04_final.sh r04h
