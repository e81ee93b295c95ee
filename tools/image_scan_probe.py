import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import semtools_amd as smt
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
rows = 10_000_000
x = torch.empty(rows, 256, device=dev)
for b in range(0, rows, 2_000_000):
    c = torch.randn(2_000_000, 256, device=dev, generator=g); x[b:b+2_000_000] = c / c.norm(dim=1, keepdim=True)
q = torch.randn(8, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
x[1234567] = q[0]
ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
c = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
o_r = torch.empty(8, 10, dtype=torch.int64, device=dev); o_d = torch.empty(8, 10, dtype=torch.float64, device=dev)
def run(nq, reps=20):
    c.search_topk_device(q.data_ptr(), nq, 10, 0, o_r.data_ptr(), o_d.data_ptr()); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): c.search_topk_device(q.data_ptr(), nq, 10, 0, o_r.data_ptr(), o_d.data_ptr())
    ctx.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, o_r[:nq].clone(), o_d[:nq].clone()
base = {n: run(n) for n in (1, 2, 4)}
c.prepack()
img = {n: run(n) for n in (1, 2, 4)}
for n in (1, 2, 4):
    print(n, "queries: f32 rows %.3f ms, with image %.3f ms, same rows %s same dist %s, uncertain %d" % (base[n][0], img[n][0], bool((base[n][1] == img[n][1]).all()), bool((base[n][2] == img[n][2]).all()), ctx.uncertain_count()))
print(img[1][1][0, :3].tolist(), img[1][2][0, :3].tolist())
