#!/bin/bash
# round 4: tokenizer.json fast path + thread grain -- parity (HF-tokenizer model dir tests) and the ingest leg with both tokenizers
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_host.py -x -q 2>&1 | tail -3
timeout 300 python tools/ingest_phases.py 1000000 > gpurun_out/r04_ingest_phases.json 2> gpurun_out/r04_ingest_phases.err; tail -5 gpurun_out/r04_ingest_phases.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_ingest_phases.json"))
print("hash", round(d["lines_per_s"] / 1e6, 2), "M lines/s", d["host_phases_ms_over_3_calls"])
print("wordpiece", d.get("wordpiece_tokenizer_json"))
PY
