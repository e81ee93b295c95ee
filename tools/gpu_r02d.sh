#!/usr/bin/env bash
out=gpurun_out; mkdir -p $out
python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -6
bash tools/trace_exchange.sh
