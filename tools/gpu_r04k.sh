#!/bin/bash
# round 4: host layer on line views -- host / store / sharded-store / CLI parity, then where an ingest call's time goes
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_host.py tests/test_gpu_sharded_store.py tests/test_gpu_embed.py -x -q 2>&1 | tail -4
timeout 300 python tools/ingest_phases.py 1000000 > gpurun_out/r04_ingest_phases.json 2> gpurun_out/r04_ingest_phases.err; tail -25 gpurun_out/r04_ingest_phases.json
