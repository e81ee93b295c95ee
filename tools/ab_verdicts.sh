#!/usr/bin/env bash
# A/B on ONE lease: the headline loop with and without the per-step verdict word (smt_search_topk_device_ex vs the plain entry point).
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
S="--steps 2000 --warmup 200 --no-secondary --no-ivfpq --no-embed --no-cpu-baseline --no-workspace --no-ingest --no-group-issue --no-c4"
for i in 1 2 3; do
  for v in 0 1; do
    SEMTOOLS_BENCH_NO_VERDICTS=$v python bench.py $S --detail-out gpurun_out/ab_verdicts_detail.json 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('no_verdicts=$v', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], d['roofline']['frac'], d.get('clk_c2_mhz'))"
  done
done
