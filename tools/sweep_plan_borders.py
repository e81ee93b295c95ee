"""K3's level plan changes shape at 2048 tiles (65 536 rows) and 131 072 tiles (4.19 M rows): ms per host-form top-10 call just below
and above each border, 64 / 256 / 1000 queries, f32 rows and image.  A jump across a border that the rows do not explain marks a plan
rule set in the wrong place."""
import gc, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import semtools_amd as smt
from semtools_amd import _lib as L
gc.disable()
dev = torch.device("cuda", 0)
ctx = smt.Context(0)
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.empty((6_000_000, 256), device=dev)
for b in range(0, 6_000_000, 2_000_000):
    c = torch.randn(2_000_000, 256, device=dev, generator=g); c /= c.norm(dim=1, keepdim=True); x[b:b + 2_000_000] = c
g.manual_seed(5)
q = torch.randn(1000, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
qh = np.ascontiguousarray(q.cpu().numpy())
o_rows = np.empty((1000, 10), dtype=np.uint64); o_dist = np.empty((1000, 10), dtype=np.float64); o_cnt = np.zeros(1000, dtype=np.uint64)
out = {}
sizes = (30_000, 60_000, 65_536, 65_600, 70_000, 100_000, 131_072, 200_000, 500_000, 1_000_000, 3_000_000, 4_000_000, 4_194_304, 4_194_400, 4_400_000, 5_000_000, 6_000_000)
for image in (False, True):
    for n in sizes:
        corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=n)
        if image: corpus.prepack()
        row = []
        for nq in (64, 256, 1000):
            def call():
                L.check(L.lib().smt_search(corpus._h, L.np_ptr(qh), nq, 10, float("nan"), smt.MODE_DOCUMENTS, None, 0, 0, L.np_ptr(o_rows), L.np_ptr(o_dist), L.np_ptr(o_cnt), 10))
            call(); call(); ctx.uncertain_count()
            t0 = time.perf_counter()
            for _ in range(5): call()
            row.append((round((time.perf_counter() - t0) / 5 * 1e3, 3), int(ctx.uncertain_count())))
        out[f"image={int(image)} rows={n}"] = row
        print(f"image={int(image)} rows={n:>8}", row, "ns per row (1000 q):", round(row[2][0] / n * 1e6, 3), file=sys.stderr)
        corpus.close()
print(json.dumps(out, indent=1))
