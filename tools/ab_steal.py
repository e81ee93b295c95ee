"""A/B of K2's dynamic groups (tuning key scan_steal = rounds per group, 0 = the static deal) in ONE process on one box: alternating
settings, the steady-state kernel time by HIP events (every launch bracketed) and the wall time per pipelined step (async select, no
events), plus the answers compared with the static deal's.  python tools/ab_steal.py [rows] > gpurun_out/r06_ab_steal.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import semtools_amd as smt
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn(rows, 256, device=dev, generator=g); x /= x.norm(dim=1, keepdim=True)
x2 = torch.randn(rows, 256, device=dev, generator=g); x2 /= x2.norm(dim=1, keepdim=True)
q = torch.randn(16, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
corpora = [smt.Corpus(ctx, device_ptr=t.data_ptr(), rows=rows) for t in (x, x2)]
out = torch.empty((64, 2, 10), dtype=torch.int64, device=dev)
st = torch.zeros(64, dtype=torch.int32, device=dev)
def run(n):
    for i in range(n):
        corpora[i & 1].search_topk_device(q[i % 16].data_ptr(), 1, 10, 0, out[i % 64, 0].data_ptr(), out[i % 64, 1].data_ptr(), out_status_ptr=st[i % 64:].data_ptr())
    ctx.synchronize()
ctx.set_tuning("prof_select", 0)
res, ref = {}, None
settings = [(0, 6), (4, 6), (0, 6), (2, 6), (0, 6), (4, 3), (0, 6), (4, 12), (0, 6), (8, 10), (0, 6), (2, 3), (0, 6), (1, 3)]
for s, pct in settings:
    ctx.set_tuning("scan_steal", s)
    ctx.set_tuning("scan_steal_pct", pct)
    ctx.set_tuning("async_select", 0)
    run(64)
    ans = out.cpu().numpy().copy()
    if ref is None: ref = ans
    same = bool((ans == ref).all()) and int(st.cpu().abs().sum()) == 0
    run(300)
    ctx.prof_enable(True); ctx.prof_reset(); run(400); n, ms = ctx.prof_read("scan"); ctx.prof_enable(False)
    ctx.set_tuning("async_select", 1)
    run(300)
    t0 = time.perf_counter(); run(1000); step = (time.perf_counter() - t0) / 1000 * 1e6
    ctx.set_tuning("async_select", 0)
    res.setdefault(f"{s}/{pct}", []).append({"scan_kernel_us": round(ms / n * 1e3, 2), "pipelined_step_us": round(step, 2), "answers_match": same})
    print(json.dumps({"scan_steal": s, "pct": pct, **res[f"{s}/{pct}"][-1]}), file=sys.stderr, flush=True)
print(json.dumps({"rows": rows, "by_scan_steal": res}, indent=1))
