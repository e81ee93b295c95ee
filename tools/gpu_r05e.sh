#!/usr/bin/env bash
# round 5: write-ahead persistence of the workspace store: host tests, then the CLI end-to-end timings with and without it
out="$(pwd)/gpurun_out"; mkdir -p "$out"
timeout 1500 python -m pytest tests/test_gpu_host.py tests/test_gpu_sharded_store.py -x -q -m gpu 2>&1 | tail -8
for wa in 1 0; do
  SEMTOOLS_WRITE_AHEAD=$wa timeout 600 python tools/bench_cli.py > "$out/r05e_cli_wa$wa.json" 2> "$out/r05e_cli_wa$wa.err"
  python - "$out/r05e_cli_wa$wa.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for name in ("workspace_cold", "workspace_warm"):
    c = d["cases"].get(name, {})
    print(sys.argv[1][-8:-5], name, "wall_s", c.get("wall_s"), {k: round(v, 1) for k, v in c.get("phases_ms", {}).items() if "persist" in k or "embed" in k or "ahead" in k or "after" in k})
PY
done
