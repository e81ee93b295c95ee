import os, sys, time, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt
dev = torch.device("cuda:0")
q = np.random.default_rng(0).standard_normal((16, 256)).astype(np.float32)
for rows in ([int(a) for a in sys.argv[1:]] or [20_000, 50_000, 100_000, 200_000, 500_000]):
    g = torch.Generator(device=dev); g.manual_seed(3)
    x = torch.randn(rows, 256, device=dev, generator=g); x /= x.norm(dim=1, keepdim=True)
    ctx = smt.Context(0)
    c = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
    for name, kw in (("plain", dict(top_k=10)), ("ranges", dict(top_k=10, ranges=[(0, rows // 2), (rows // 2 + 1, rows)]))):
        for i in range(5):
            c.search(q[i], **kw)
        ctx.prof_enable(True); ctx.prof_reset()
        t0 = time.perf_counter()
        for i in range(30):
            c.search(q[i % 16], **kw)
        dt = (time.perf_counter() - t0) / 30
        ns, ms = ctx.prof_read("scan"); n2, ms2 = ctx.prof_read("select")
        ctx.prof_enable(False)
        print(json.dumps(dict(rows=rows, mode=name, us=round(dt * 1e6, 1), scan_us=round(ms / max(ns, 1) * 1e3, 1), select_us=round(ms2 / max(n2, 1) * 1e3, 1))), flush=True)
    c.close(); ctx.close()
