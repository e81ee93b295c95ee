#!/usr/bin/env bash
# A/B: who enqueues the n - 1 waits in front of the merge of a one-process group search (8 logical shards): the ranks' issuing threads
# (default) or the caller's thread ($SEMTOOLS_GROUP_WAITS=caller).  Median of five rounds each, three times alternating.
for rep in 1 2 3; do for w in issuers caller; do
SEMTOOLS_GROUP_WAITS=$w python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
import semtools_amd as smt
out = []
for n in (4, 8):
    r = bench.bench_group_issue(smt, torch.device("cuda", 0), n_shards=n)
    out.append((n, round(r["host_issue_us_per_search"], 1), round(r["every_rank_wants_the_answer_us"], 1), r["checks"]["last_answer_matches_fp64_topk"]))
print("waits by", os.environ["SEMTOOLS_GROUP_WAITS"], out)
PY
done; done
