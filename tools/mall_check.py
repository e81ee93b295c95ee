"""Does the 256 MiB Infinity Cache (MALL) flatter the 1 M-row scan when the SAME 1 GB shard is scanned every step?
Steady-state K2 kernel time with 1, 2 and 4 rotating copies of the corpus (rotate >= 2 copies)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt
rows = 1_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
shards = []
for c in range(4):
    g.manual_seed(3 + 100 * c)
    x = torch.randn(rows, 256, device=dev, generator=g); x /= x.norm(dim=1, keepdim=True)
    shards.append(x)
q = torch.randn(16, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
torch.cuda.synchronize()
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
corp = [smt.Corpus(ctx, device_ptr=s.data_ptr(), rows=rows) for s in shards]
out_r = torch.empty(10, dtype=torch.int64, device=dev); out_d = torch.empty(10, dtype=torch.float64, device=dev)
def run(n, copies):
    for i in range(n):
        corp[i % copies].search_topk_device(q[i % 16].data_ptr(), 1, 10, 0, out_r.data_ptr(), out_d.data_ptr())
    torch.cuda.synchronize()
run(800, 1)
ctx.set_tuning("prof_select", 0)
for rnd in range(2):
    for copies in (1, 2, 4):
        run(100, copies)
        ctx.prof_enable(True); ctx.prof_reset()
        run(400, copies)
        n, ms = ctx.prof_read("scan")
        ctx.prof_enable(False)
        print(json.dumps(dict(copies=copies, scan_us=round(ms / n * 1e3, 2))))
