"""One GPU, 100 M rows (c4's corpus): 1 / 4 / 16 / 64 queries through the scan kernel over the f32 rows and through the batched
kernel over the fp16 operand image (forced f16 x 2 / f16 x 1: the auto rule stops fp16 nominations at 32 M rows per shard): time,
agreement, queries without an exactness certificate."""
import json, sys, time
import torch
sys.path.insert(0, "/root/repo")
import semtools_amd as smt
dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.empty(rows, 256, device=dev)
for b in range(0, rows, 4_000_000):
    n = min(4_000_000, rows - b)
    c = torch.randn(n, 256, device=dev, generator=g); x[b:b + n] = c / c.norm(dim=1, keepdim=True); del c
g.manual_seed(4)
q = torch.randn(64, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
c = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
o_r = torch.empty(64, 10, dtype=torch.int64, device=dev); o_d = torch.empty(64, 10, dtype=torch.float64, device=dev)
def run(nq, reps=5):
    c.search_topk_device(q.data_ptr(), nq, 10, 0, o_r.data_ptr(), o_d.data_ptr()); ctx.synchronize(); ctx.uncertain_count()
    t0 = time.perf_counter()
    for _ in range(reps): c.search_topk_device(q.data_ptr(), nq, 10, 0, o_r.data_ptr(), o_d.data_ptr())
    ctx.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, o_r[:nq].clone(), o_d[:nq].clone(), ctx.uncertain_count() // reps
out = {"rows": rows}
ctx.set_tuning("image_scan_min_rows", 0)
base = {n: run(n) for n in (1, 4, 16, 64)}
t0 = time.perf_counter(); c.prepack(); ctx.synchronize(); out["prepack_ms"] = (time.perf_counter() - t0) * 1e3
ctx.set_tuning("image_scan_min_rows", 4_000_000)
for mode, name in ((2, "f16x2"), (3, "f16x1")):
    ctx.set_tuning("gemm_nominate", mode)
    for n in (1, 4, 16, 64):
        ms, r, d, unc = run(n)
        out[f"{name}_{n}q"] = {"ms": round(ms, 3), "f32_path_ms": round(base[n][0], 3), "same": bool((r == base[n][1]).all()) and bool((d == base[n][2]).all()), "uncertain_per_call": unc}
ctx.set_tuning("gemm_nominate", 0)
print(json.dumps(out))
