#!/usr/bin/env bash
out=gpurun_out; mkdir -p $out
python -m pytest tests/test_gpu_ivfpq.py tests/test_gpu_host.py -m gpu -q -x --timeout 900 2>&1 | tail -15
