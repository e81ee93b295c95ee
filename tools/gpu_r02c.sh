#!/usr/bin/env bash
out=gpurun_out; mkdir -p $out
python -m pytest tests/test_gpu_batched.py tests/test_gpu_fuzz.py -m gpu -q -x --timeout 600 2>&1 | tail -3
python tools/bench_small_batch.py --nq 8 32 64 96 128 --variants 1 2>&1 | grep '"nq"'
python tools/bench_batched.py --nq 1000 --rows 10000000 2>&1 | grep rows
