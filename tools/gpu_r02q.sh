#!/usr/bin/env bash
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_nearties.py -x -q 2>&1 | tail -3
for rep in 1 2; do
  echo "--- new"
  timeout 600 python tools/bench_small_batch.py --nq 32 128 160 256 512 1000 --variants 1 --reps 5 2>&1 | grep -E "^\{" | cut -c1-175
done
