"""Fine sweep of waves/CU and unroll for K2 at 1 M rows, steady state (settle first, interleaved rounds)."""
import os, sys, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt

rows = 1_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn(rows, 256, device=dev, generator=g); x /= x.norm(dim=1, keepdim=True)
q = torch.randn(16, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
out_r = torch.empty(10, dtype=torch.int64, device=dev); out_d = torch.empty(10, dtype=torch.float64, device=dev)
def run(n):
    for i in range(n):
        corpus.search_topk_device(q[i % 16].data_ptr(), 1, 10, 0, out_r.data_ptr(), out_d.data_ptr())
    torch.cuda.synchronize()
run(600)
configs = [(t, u, b) for t in (384, 448, 512, 576, 640) for u in (2, 4) for b in (256,)] + [(256, 4, 512), (256, 8, 512), (320, 4, 512), (512, 4, 128)]
res = {c: [] for c in configs}
ctx.set_tuning("prof_select", 0)
for rnd in range(4):
    for c in configs:
        t, u, b = c
        ctx.set_tuning("scan_threads", t); ctx.set_tuning("scan_unroll", u); ctx.set_tuning("scan_blocks", b)
        run(30)
        ctx.prof_enable(True); ctx.prof_reset()
        run(300)
        n, ms = ctx.prof_read("scan")
        ctx.prof_enable(False)
        res[c].append(ms / n * 1e3)
for c in configs:
    v = res[c]
    print(json.dumps(dict(threads=c[0], unroll=c[1], blocks=c[2], waves_per_cu=c[0] // 64 * c[2] // 256, us_median=round(float(np.median(v)), 1), us_all=[round(a, 1) for a in v])))
