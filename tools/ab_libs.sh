#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by more than most effects): semtools_amd/lib/ab/libA.so and libB.so are
# copied over the product library in turn, bench.py runs the headline leg only, three rounds each; prints kernel us / step us per run.
set -u
mkdir -p gpurun_out
LIB=semtools_amd/lib/libsemtools_hip.so
cp $LIB /tmp/lib_orig.so
for round in 1 2 3; do
  for v in A B; do
    cp semtools_amd/lib/ab/lib$v.so $LIB
    timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary --no-embed --no-ivfpq --no-c4 --no-group-issue --no-workspace --no-ingest --no-small-calls \
        --detail-out gpurun_out/ab_$v$round.json 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('$v', $round, 'ms_per_step', round(d['ms_per_step'],5), 'kernel_us', r.get('kernel_us') or r.get('avg_kernel_us') or r, 'frac', r['frac'])"
  done
done
cp /tmp/lib_orig.so $LIB
