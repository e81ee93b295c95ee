#!/bin/bash
# round 4: K1 with 32-bit run-relative positions (no spills) -- parity + A/B against the round-3 kernel in one process
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_embed.py -x -q 2>&1 | tail -3
timeout 300 python tools/embed_ab.py 1 3 > gpurun_out/embed_ab_r04j.json 2> gpurun_out/embed_ab_r04j.err; python -c "import json; d=json.load(open(\"gpurun_out/embed_ab_r04j.json\")); [print(k, d[k]) for k in d if k.startswith(\"embed_\")]"
