#!/usr/bin/env bash
# bf16 x 3 nomination: parity first, then A/B against the f32 MFMA path inside one box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_nearties.py -x -q -s 2>&1 | grep -E "max \||passed|failed|Error|error" | tail -15
for bf in 1 0; do
  echo "--- gemm_bf16x3=$bf"
  timeout 600 python tools/bench_small_batch.py --nq 8 32 64 96 128 256 1000 --variants 1 --reps 5 --tune gemm_bf16x3=$bf 2>&1 | grep -E "^\{" 
done
