#!/usr/bin/env bash
# round 2, session B: the LDS-row small-batch kernel -- parity first, then timing against the resident kernel
out=gpurun_out; mkdir -p $out
python -m pytest tests/test_gpu_batched.py tests/test_gpu_fuzz.py tests/test_gpu_nearties.py tests/test_gpu_group.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 2>&1 | tail -30 > $out/pytest_r02b.log; tail -12 $out/pytest_r02b.log
timeout 600 python tools/bench_small_batch.py --rows 10000000 --nq 8 16 32 64 96 128 --ranges --out $out/k3_small_r02b.json 2>&1 | tail -30
