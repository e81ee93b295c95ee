"""Where one ingest call's wall time goes (bench.py's ingest leg alone): the host layer's phase timer, cumulative over the three calls."""
import json
import sys

import torch

sys.path.insert(0, "/root/repo")
import bench  # noqa: E402
import semtools_amd as smt  # noqa: E402

ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
r = bench.bench_ingest(smt, ctx, n)
r["sum_of_phases_ms"] = round(sum(v for k, v in (r["host_phases_ms_over_3_calls"] or {}).items() if not k.startswith("within_")), 1)
print(json.dumps(r, indent=1))
