#!/usr/bin/env bash
# A/B inside one box: working tree ("new") against the copy under ab_old/ ("old"), alternating; args = bench_small_batch flags
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"
for rep in 1 2 3; do
  for side in new old; do
    if [ $side = new ]; then tool="$root/tools/bench_small_batch.py"; else tool="$root/ab_old/tools/bench_small_batch.py"; fi
    echo "--- $side"
    timeout 600 python "$tool" "$@" 2>&1 | grep -E "^\{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['nq'], 'gemm_ms', d['gemm_ms'], 'wall_ms', d['wall_ms'], d['k2_agreement'])"
  done
done
