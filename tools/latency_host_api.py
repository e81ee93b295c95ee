"""End-to-end latency of the HOST entry point smt_search (host query in, host results out) -- what a CLI call pays."""
import os, sys, time, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt

dev = torch.device("cuda:0")
for rows in (1000, 100_000, 1_000_000):
    g = torch.Generator(device=dev); g.manual_seed(3)
    x = torch.randn(rows, 256, device=dev, generator=g); x /= x.norm(dim=1, keepdim=True)
    q = np.random.default_rng(0).standard_normal((16, 256)).astype(np.float32)
    ctx = smt.Context(0)
    c = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
    for mode, kw in (("top10", dict(top_k=10)), ("top10+ranges", dict(top_k=10, ranges=[(0, rows // 2), (rows // 2 + 1, rows)])),
                     ("thr", dict(max_distance=0.75))):
        for i in range(5):
            c.search(q[i], **kw)
        t0 = time.perf_counter()
        n = 50
        for i in range(n):
            c.search(q[i % 16], **kw)
        dt = (time.perf_counter() - t0) / n
        print(json.dumps(dict(rows=rows, mode=mode, us_per_call=round(dt * 1e6, 1))), flush=True)
    c.close(); ctx.close()
