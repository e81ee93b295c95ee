#!/usr/bin/env bash
# round 2, session A: full GPU test suite, default bench, bench with the exchange forced on one rank (library RCCL path)
out=gpurun_out; mkdir -p $out
python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -40 > $out/pytest_r02a.log; tail -15 $out/pytest_r02a.log
timeout 900 python bench.py > $out/bench_r02a.json 2> $out/bench_r02a.err; cat $out/bench_r02a.json; tail -5 $out/bench_r02a.err
SEMTOOLS_BENCH_FORCE_EXCHANGE=1 timeout 600 python bench.py --steps 1000 --warmup 100 --no-secondary --no-ivfpq --no-cpu-baseline --c4-rows 20000000 > $out/bench_r02a_forced.json 2> $out/bench_r02a_forced.err; cat $out/bench_r02a_forced.json; tail -5 $out/bench_r02a_forced.err
