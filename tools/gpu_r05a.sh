#!/usr/bin/env bash
# round 5, first GPU call: the peer transport of one-process groups and bench.py's one-process launcher
#   group / sharded-store / defaults tests, then the group-issue figures (peer vs copy transport, 1/2/4/8 logical shards)
tag="${1:-r05a}"
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_gpu_sharded_store.py -x -q -m gpu 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_gpu_defaults.py -x -q -m gpu -k "bench" 2>&1 | tail -15
timeout 600 python - <<'PY' > "$out/${tag}_group_issue.json" 2> "$out/${tag}_group_issue.err"
import json, sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
import semtools_amd as smt
res = {}
for n in (1, 2, 4, 8):
    r = bench.bench_group_issue(smt, torch.device("cuda", 0), n_shards=n)
    res[str(n)] = {k: r[k] for k in ("host_issue_us_per_search", "end_to_end_us_per_search", "every_rank_wants_the_answer_us", "copy_transport_us",
                                     "copy_transport_every_rank_us", "one_thread_issues_every_shard_us", "checks")}
print(json.dumps(res, indent=1))
PY
echo "group issue rc=$?"; cat "$out/${tag}_group_issue.json"; tail -5 "$out/${tag}_group_issue.err"
