#!/usr/bin/env bash
# HIP API calls of one 8-shard logical-group search (one caller thread, one answer): count and host time per call, with and without
# the issuing threads.  rocprofv3 --hip-trace --stats only (no counters).
tag="${1:-r05}"
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
cat > /tmp/gi.py <<'PY'
import ctypes as C, os, sys, time, json
sys.path.insert(0, os.environ["ROOT"])
import torch
import semtools_amd as smt
from semtools_amd import _lib as L
n_shards, rows, k, n = int(os.environ.get("SHARDS", 8)), 1_000_000, 10, int(os.environ.get("N", 400))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
shards = []
for _ in range(n_shards):
    x = torch.randn(rows, 256, device=dev, generator=g); x /= x.norm(dim=1, keepdim=True); shards.append(x)
q = torch.randn(16, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
torch.cuda.synchronize()
grp = smt.Group.logical(0, n_shards)
if os.environ.get("TRANSPORT"): grp.set_transport(os.environ["TRANSPORT"])
sc = smt.ShardedCorpus(grp, device_ptrs=[s.data_ptr() for s in shards], shard_rows=[rows] * n_shards)
for i in range(n_shards): grp.ctx(i).set_tuning("async_select", int(os.environ.get("ASYNC", "1")))
out0 = torch.empty((2, k), dtype=torch.int64, device=dev)
qp = [(C.c_void_p * n_shards)(*[C.c_void_p(q[j].data_ptr())] * n_shards) for j in range(16)]
op = (C.c_void_p * n_shards)(*([C.c_void_p(out0.data_ptr())] + [C.c_void_p(None)] * (n_shards - 1)))
fn = L.lib().smt_sharded_search_topk_device
for j in range(8): L.check(fn(sc._h, qp[j % 16], 1, k, op))
grp.synchronize()
ts = []
t0 = time.perf_counter()
for j in range(n):
    a = time.perf_counter(); fn(sc._h, qp[j % 16], 1, k, op); ts.append(time.perf_counter() - a)
issued = time.perf_counter() - t0
grp.synchronize()
ts.sort()
print(json.dumps({"shards": n_shards, "threads": os.environ.get("SEMTOOLS_GROUP_THREADS", "1"), "transport": grp.transport, "n": n,
                  "issue_us_avg": issued / n * 1e6, "issue_us_median": ts[len(ts) // 2] * 1e6, "issue_us_p10": ts[len(ts) // 10] * 1e6,
                  "first_40_avg_us": None}))
PY
export ROOT="$root"
cd /tmp && export TMPDIR=/tmp
for th in 1 0; do
  echo "== plain run, SEMTOOLS_GROUP_THREADS=$th"
  SEMTOOLS_GROUP_THREADS=$th N=40 python /tmp/gi.py 2>/dev/null | tail -1
  SEMTOOLS_GROUP_THREADS=$th N=400 python /tmp/gi.py 2>/dev/null | tail -1
done
for th in 1 0; do
  SEMTOOLS_GROUP_THREADS=$th N=40 rocprofv3 --hip-trace --stats --output-format csv -d "$out/trace_gi_$th" -o t -- python /tmp/gi.py > "$out/trace_gi_$th.log" 2>&1
  f=$(find "$out/trace_gi_$th" -name "*hip_api_stats.csv" | head -1)
  echo "== hip api stats, threads=$th ($f)"; head -14 "$f" | cut -c1-140
  cp "$f" "$out/${tag}_group_issue_hip_api_stats_threads$th.csv" 2>/dev/null
done
find "$out" -name "*.csv" -size +4M -delete
