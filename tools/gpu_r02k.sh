#!/usr/bin/env bash
out=gpurun_out; mkdir -p $out
python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -5
python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python bench.py > $out/bench_r02k.json 2> $out/bench_r02k.err; cat $out/bench_r02k.json | cut -c1-6000; tail -3 $out/bench_r02k.err
