"""Interleaved A/B rounds of K2 configurations (median/min of scan kernel time over rounds)."""
import json, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt

dev = torch.device("cuda:0")
rows, k = int(os.environ.get("ROWS", 1_000_000)), 10
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.empty(rows, 256, device=dev)
for b in range(0, rows, 4_000_000):  # chunked: a 100M-row corpus (102 GB) is generated without a second copy
    e = min(rows, b + 4_000_000)
    t = torch.randn(e - b, 256, device=dev, generator=g); x[b:e] = t / t.norm(dim=1, keepdim=True)
del t
q = torch.randn(4, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
out_rows = torch.empty(4, k, dtype=torch.int64, device=dev); out_dist = torch.empty(4, k, dtype=torch.float64, device=dev)
ctx = smt.Context(0); corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows); ctx.prof_enable(True)
configs = [tuple(map(int, c.split(","))) for c in sys.argv[1:]]  # threads,blocks,unroll,prefetch,nt
res = {c: [] for c in configs}
for rnd in range(int(os.environ.get('ROUNDS', 7))):
    for c in configs:
        th, bl, un, pf, nt = c
        ctx.set_tuning("scan_threads", th); ctx.set_tuning("scan_blocks", bl); ctx.set_tuning("scan_unroll", un)
        ctx.set_tuning("scan_prefetch", pf); ctx.set_tuning("scan_nontemporal", nt)
        for _ in range(3):
            corpus.search_topk_device(q.data_ptr(), 1, k, 0, out_rows.data_ptr(), out_dist.data_ptr())
        ctx.synchronize(); ctx.prof_reset()
        for _ in range(int(os.environ.get('REPS', 40))):
            corpus.search_topk_device(q.data_ptr(), 1, k, 0, out_rows.data_ptr(), out_dist.data_ptr())
        n, ms = ctx.prof_read("scan")
        res[c].append(ms / n * 1e3)
for c in configs:
    v = res[c]
    print(json.dumps(dict(threads=c[0], blocks=c[1], unroll=c[2], prefetch=c[3], nt=c[4], med_us=round(statistics.median(v), 2),
                          min_us=round(min(v), 2), max_us=round(max(v), 2), med_GBps=round(rows * 1024 / statistics.median(v) / 1e3, 1))))
