#!/usr/bin/env bash
# A/B on THIS box: issuing threads and the caller spin before they block ($SEMTOOLS_GROUP_SPIN_US, default 100) against blocking at
# once (0).  Boxes differ (host idle states): run it on several leases.
for rep in 1 2; do for sp in 100 0; do
SEMTOOLS_GROUP_SPIN_US=$sp python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
import semtools_amd as smt
out = []
for n in (4, 8):
    r = bench.bench_group_issue(smt, torch.device("cuda", 0), n_shards=n)
    out.append((n, round(r["host_issue_us_per_search"], 1), round(r["every_rank_wants_the_answer_us"], 1), r["checks"]["last_answer_matches_fp64_topk"]))
print("spin us", os.environ["SEMTOOLS_GROUP_SPIN_US"], out)
PY
done; done
