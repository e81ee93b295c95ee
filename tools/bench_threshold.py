"""K4 (threshold mode) scan rate at 1 M rows: smt_search with max_distance through the host ABI; the scan kernel
time comes from the library's HIP events ("scan"), the rest (rescoring, D2H of the hits, host sort) is in wall."""
import os, sys, time, json
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn(rows, 256, device=dev, generator=g); x /= x.norm(dim=1, keepdim=True)
q = torch.randn(256, device=dev, generator=g); q /= q.norm()
qh = q.cpu().numpy()
ctx = smt.Context(0)
corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
for tune in sys.argv[2:]:
    k, v = tune.split("="); ctx.set_tuning(k, int(v))
for md in (0.5, 0.8, 0.9):
    for _ in range(30):
        corpus.search(qh, max_distance=md)
    ctx.prof_enable(True); ctx.prof_reset()
    t0 = time.perf_counter(); reps = 50
    for _ in range(reps):
        hits = corpus.search(qh, max_distance=md)[0]
    wall = (time.perf_counter() - t0) / reps
    n, ms = ctx.prof_read("scan")
    ctx.prof_enable(False)
    print(json.dumps(dict(max_distance=md, hits=len(hits[0]), scan_us=round(ms / n * 1e3, 1), scan_TBps=round(rows * 1024 / (ms / n * 1e-3) / 1e12, 2),
                          wall_us=round(wall * 1e6, 1))))
