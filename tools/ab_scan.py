"""A/B of two builds of the library (one gpurun call = one box; boxes differ by more than the 1-2 % under test).
Setup: git worktree add -f ab_old <rev> && bash ab_old/semtools_amd/csrc/build.sh  (ab_old/ is git-ignored, travels with gpurun).
Run once per build root (argv[1]) inside ONE gpurun call, alternating, and compare the steady-state K2 kernel time."""
import os, sys, json
root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root)
import numpy as np
import torch
import semtools_amd as smt
assert os.path.abspath(smt.__file__).startswith(root), smt.__file__
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
for kv in sys.argv[2:]:
    k, val = kv.split("="); ctx.set_tuning(k, int(val))
run(800)
ctx.set_tuning("prof_select", 0)
v = []
for rnd in range(5):
    ctx.prof_enable(True); ctx.prof_reset()
    run(400)
    n, ms = ctx.prof_read("scan")
    ctx.prof_enable(False)
    v.append(round(ms / n * 1e3, 2))
print(json.dumps(dict(root=os.path.basename(root) or "repo", tune=sys.argv[2:], scan_us=v)))
