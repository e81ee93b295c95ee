"""Where the routes cross, measured route by route (tuning keys force each): ms per host-form top-10 call, 1..7 queries, 50 k .. 2 M rows:
K2 (scan kernel over f32 rows), K3 over f32 rows (nq >= 2), K3 over the operand image.  python tools/sweep_crossover.py > gpurun_out/sweep_crossover.json"""
import gc, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import semtools_amd as smt
import ctypes as C
from semtools_amd import _lib as L

gc.disable()   # (a full collection with torch imported takes 30-45 ms and lands in the middle of a timing loop)
dev = torch.device("cuda", 0)
ctx = smt.Context(0)
g = torch.Generator(device=dev); g.manual_seed(3)
rows = 2_000_000
x = torch.randn(rows, 256, device=dev, generator=g); x /= x.norm(dim=1, keepdim=True)
g.manual_seed(4)
q = torch.randn(8, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
qh = np.ascontiguousarray(q.cpu().numpy())
o_rows = np.empty((8, 10), dtype=np.uint64); o_dist = np.empty((8, 10), dtype=np.float64); o_cnt = np.zeros(8, dtype=np.uint64)
sizes = (1000, 50_000, 100_000, 200_000, 300_000, 500_000, 700_000, 1_000_000, 1_500_000, 2_000_000)
out = {}
for route in ("k2", "k3_f32", "k3_image"):
    for n in sizes:
        corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=n)
        if route == "k3_image": corpus.prepack()
        ctx.set_tuning("image_scan_min_rows", 1 if route == "k3_image" else 0)
        ctx.set_tuning("gemm_min_rows_small", 0 if route != "k2" else 1 << 40)
        ctx.set_tuning("gemm_min_nq", 2 if route != "k2" else 8)
        row = []
        for nq in range(1, 8):
            if route == "k3_f32" and nq == 1: row.append(None); continue
            def call():
                L.check(L.lib().smt_search(corpus._h, L.np_ptr(qh), nq, 10, float("nan"), smt.MODE_DOCUMENTS, None, 0, 0,
                                           L.np_ptr(o_rows), L.np_ptr(o_dist), L.np_ptr(o_cnt), 10))
            call(); call(); ctx.synchronize()
            best = []
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(20): call()
                best.append((time.perf_counter() - t0) / 20 * 1e6)
            row.append(round(float(np.median(best)), 1))
        out[f"{route} rows={n}"] = row
        print(f"{route:9} {n:>8}", row, file=sys.stderr)
        corpus.close()
print(json.dumps(out, indent=1))
