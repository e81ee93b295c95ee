"""Which nomination mode for 33 .. 320 queries: ms per host-form top-10 call with gemm_nominate forced to 1 (bf16 x 3), 2 (f16 x 2),
3 (f16 x 1) and 0 (auto), f32 rows and operand image, 2 M and 10 M rows; re-answered queries in brackets."""
import gc, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import semtools_amd as smt
from semtools_amd import _lib as L
gc.disable()
dev = torch.device("cuda", 0)
ctx = smt.Context(0)
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.empty((10_000_000, 256), device=dev)
for b in range(0, 10_000_000, 2_000_000):
    c = torch.randn(2_000_000, 256, device=dev, generator=g); c /= c.norm(dim=1, keepdim=True); x[b:b + 2_000_000] = c
g.manual_seed(5)
q = torch.randn(1000, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
qh = np.ascontiguousarray(q.cpu().numpy())
o_rows = np.empty((1000, 10), dtype=np.uint64); o_dist = np.empty((1000, 10), dtype=np.float64); o_cnt = np.zeros(1000, dtype=np.uint64)
out = {}
for image in (False, True):
    for n in (2_000_000, 10_000_000):
        corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=n)
        if image: corpus.prepack()
        for nq in (33, 64, 96, 128, 160, 200, 224, 256, 320):
            row = {}
            for mode in (0, 1, 2, 3):
                ctx.set_tuning("gemm_nominate", mode)
                def call():
                    L.check(L.lib().smt_search(corpus._h, L.np_ptr(qh), nq, 10, float("nan"), smt.MODE_DOCUMENTS, None, 0, 0, L.np_ptr(o_rows), L.np_ptr(o_dist), L.np_ptr(o_cnt), 10))
                call(); call(); ctx.uncertain_count()
                t0 = time.perf_counter()
                for _ in range(5): call()
                row[("auto", "bf16x3", "f16x2", "f16x1")[mode]] = (round((time.perf_counter() - t0) / 5 * 1e3, 3), int(ctx.uncertain_count()))
            ctx.set_tuning("gemm_nominate", 0)
            out[f"image={int(image)} rows={n} nq={nq}"] = row
            print(f"image={int(image)} rows={n} nq={nq}", row, file=sys.stderr)
        corpus.close()
print(json.dumps(out, indent=1))
