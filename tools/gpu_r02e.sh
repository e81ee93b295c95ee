#!/usr/bin/env bash
out=gpurun_out; mkdir -p $out
python -m pytest tests/test_gpu_ivfpq.py -m gpu -q -x --timeout 900 2>&1 | tail -8
timeout 900 python tools/sweep_ivf.py --rows 10000000 --nprobe 1 4 8 16 32 --rerank 32 64 128 512 --out $out/ivf_sweep_r02e.json 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Lib"
