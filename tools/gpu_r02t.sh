#!/usr/bin/env bash
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"
for rows in 10000000 2000000 1000000 300000; do
for mn in 5 8; do
  echo "--- rows $rows gemm_min_nq=$mn"
  timeout 300 python tools/bench_small_batch.py --rows $rows --nq 5 6 7 --variants 1 --reps 9 --tune gemm_min_nq=$mn --tune gemm_min_rows_small=0 2>&1 | grep -E "^\{" | python -c "
import sys, json
print('  '.join('%d: wall %.3f %s' % (json.loads(l)['nq'], json.loads(l)['wall_ms'], json.loads(l)['k2_agreement']) for l in sys.stdin))"
done
done
