for th in 0 2 3 4 1; do
SEMTOOLS_GROUP_THREADS=$th python - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench
import semtools_amd as smt
r = bench.bench_group_issue(smt, torch.device("cuda", 0), n_shards=8)
print("threads", os.environ["SEMTOOLS_GROUP_THREADS"], {k: round(r[k], 1) for k in ("host_issue_us_per_search", "every_rank_wants_the_answer_us", "copy_transport_us")}, r["checks"])
PY
done
