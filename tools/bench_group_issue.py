"""What ONE host thread pays to drive an 8-shard group.  The caller of the library is one synchronous process
(/root/reference/src/bin/semtools.rs:134-135); in single-process mode that thread issues scan + select + gather + merge for every
shard.  On a 1-GPU box the shards are logical ranks of one device (copy transport), so the GPU work of the 8 shards runs back to
back -- but the HOST time to issue a search is what an 8-GPU node's thread would pay too: it must stay well below one shard's scan
(~150 us at 1 M rows) or the group needs one issuing thread per device.

Reports, per search of 1 query over 8 x 1 M rows: host microseconds to ISSUE smt_sharded_search_topk_device (device-resident form;
nothing synchronises), the same with the select stage on the aux stream (async_select), and the whole smt_sharded_search host call
(host in, host out) against the GPU time it waits for."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt  # noqa: E402
from semtools_amd import _lib as L  # noqa: E402

n_shards = int(os.environ.get("SHARDS", 8))
rows = int(os.environ.get("ROWS", 1_000_000))
k = 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(3)
shards = []
for s in range(n_shards):
    x = torch.randn(rows, 256, device=dev, generator=g)
    x /= x.norm(dim=1, keepdim=True)
    shards.append(x)
q = torch.randn(16, 256, device=dev, generator=g)
q /= q.norm(dim=1, keepdim=True)
torch.cuda.synchronize()

out = {"shards": n_shards, "rows_per_shard": rows, "top_k": k, "transport": "logical ranks on one device (event-ordered copies instead of RCCL)"}
for nsh in sorted({1, 2, 4, n_shards}):
    grp = smt.Group.logical(0, nsh)
    sc = smt.ShardedCorpus(grp, device_ptrs=[sh.data_ptr() for sh in shards[:nsh]], shard_rows=[rows] * nsh)
    outs = [torch.empty((2, k), dtype=torch.int64, device=dev) for _ in range(nsh)]
    leg = {}
    for async_sel in (0, 1):
        for i in range(nsh):
            grp.ctx(i).set_tuning("async_select", async_sel)
        qp = [(C.c_void_p * nsh)(*[C.c_void_p(q[j].data_ptr())] * nsh) for j in range(16)]
        op = (C.c_void_p * nsh)(*[C.c_void_p(o.data_ptr()) for o in outs])
        fn = L.lib().smt_sharded_search_topk_device
        for j in range(8):
            L.check(fn(sc._h, qp[j % 16], 1, k, op))
        grp.synchronize()
        n = 40
        t0 = time.perf_counter()
        for j in range(n):
            fn(sc._h, qp[j % 16], 1, k, op)
        issued = time.perf_counter() - t0
        grp.synchronize()
        total = time.perf_counter() - t0
        leg["async_select" if async_sel else "in_order"] = {"host_issue_us_per_search": issued / n * 1e6, "us_per_search_end_to_end": total / n * 1e6,
                                                             "host_issue_us_per_shard": issued / n / nsh * 1e6}
    for i in range(nsh):
        grp.ctx(i).set_tuning("async_select", 0)
    # what ONE thread pays when it issues the shards' scan + select itself, one after the other, with nothing to exchange
    # (smt_search_topk_device per shard): the cost the group's issuing threads take off the caller
    views = [sc.shard(i)[0] for i in range(nsh)]
    o_r = [torch.empty(k, dtype=torch.int64, device=dev) for _ in range(nsh)]
    o_d = [torch.empty(k, dtype=torch.float64, device=dev) for _ in range(nsh)]
    fn1 = L.lib().smt_search_topk_device
    args1 = [(views[i]._h, C.c_void_p(q[0].data_ptr()), 1, k, 0, C.c_void_p(o_r[i].data_ptr()), C.c_void_p(o_d[i].data_ptr())) for i in range(nsh)]
    for a in args1:
        L.check(fn1(*a))
    grp.synchronize()
    t0 = time.perf_counter()
    for j in range(n):
        for a in args1:
            fn1(*a)
    leg["one_thread_issues_every_shard_us_per_search"] = (time.perf_counter() - t0) / n * 1e6
    grp.synchronize()
    # the host API (what the store calls): one query in, hits out
    qh = q.cpu().numpy()
    for j in range(4):
        sc.search(qh[j], top_k=k)
    n = 40
    t0 = time.perf_counter()
    for j in range(n):
        got = sc.search(qh[j % 16], top_k=k)
    host_call = (time.perf_counter() - t0) / n
    leg["host_api_smt_sharded_search_us_per_call"] = host_call * 1e6
    # correctness of the last answer against the unsharded fp64 top-k
    allx = torch.cat(shards[:nsh])
    d = 1.0 - (allx.double() @ q[(n - 1) % 16].double())
    tv, ti = torch.topk(d, k, largest=False)
    leg["last_answer_matches_fp64_topk"] = bool(got[0][0].tolist() == ti.cpu().tolist())
    del allx, d
    out[f"{nsh}_shards"] = leg
    sc.close()
    grp.close()
one_scan_us = 150.0
out["note"] = ("host_issue = the caller's thread inside smt_sharded_search_topk_device: wake the issuing threads (one per shard; each binds, "
               "launches scan + select), wait for them, then the exchange (copy transport here: ~5 n event / copy calls; RCCL on a real "
               "node: ncclGroupStart + n ncclAllGather + ncclGroupEnd) and the merge launches")
out["verdict"] = {"one_shard_scan_us": one_scan_us, "budget_us": one_scan_us / 2,
                  "host_issue_us_8_shards_in_order": out[f"{n_shards}_shards"]["in_order"]["host_issue_us_per_search"],
                  "within_budget": out[f"{n_shards}_shards"]["in_order"]["host_issue_us_per_search"] <= one_scan_us / 2}
print(json.dumps(out, indent=1))
