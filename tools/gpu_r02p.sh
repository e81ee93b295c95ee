#!/usr/bin/env bash
# row-register kernel: parity, then A/B against ab_old (previous commit) inside one box
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_nearties.py -x -q 2>&1 | tail -3
for rep in 1 2; do
  for side in new old; do
    if [ $side = new ]; then tool="$root/tools/bench_small_batch.py"; else tool="$root/ab_old/tools/bench_small_batch.py"; fi
    echo "--- $side"
    timeout 600 python "$tool" --nq 8 32 64 128 256 1000 --variants 1 --reps 5 2>&1 | grep -E "^\{" | cut -c1-175
  done
done
