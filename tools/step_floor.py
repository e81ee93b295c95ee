"""Whole-step time of the c2 loop (1 query x 1 M rows, results to pinned host) with/without HIP-event profiling and
with/without async select: where does the per-step time beyond the scan kernel go?"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt
rows, k = 1_000_000, 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn(rows, 256, device=dev, generator=g); x /= x.norm(dim=1, keepdim=True)
q = torch.randn(16, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
torch.cuda.synchronize()
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
host = torch.empty((64, 2, k), dtype=torch.int64).pin_memory()
def run(n):
    for i in range(n):
        s = host[i % 64]
        corpus.search_topk_device(q[i % 16].data_ptr(), 1, k, 0, s[0].data_ptr(), s[1].data_ptr())
    torch.cuda.synchronize()
run(800)
ctx.set_tuning("prof_select", 0)
for tune in sys.argv[1:]:
    kk, v = tune.split("="); ctx.set_tuning(kk, int(v))
for asyn in (0, 1, 0, 1):
    for prof in (0, 1):
        ctx.set_tuning("async_select", asyn)
        ctx.prof_enable(bool(prof)); ctx.prof_reset()
        run(200)
        t0 = time.perf_counter(); run(2000); dt = (time.perf_counter() - t0) / 2000
        n, ms = ctx.prof_read("scan") if prof else (0, 0.0)
        ctx.prof_enable(False)
        print(json.dumps(dict(async_select=asyn, hip_events=prof, step_us=round(dt * 1e6, 2), scan_us=round(ms / n * 1e3, 2) if n else None)))
