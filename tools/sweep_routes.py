"""Routing sanity: ms per host-form top-10 call across corpus sizes (1 k .. 30 M rows) and query counts (1 .. 64), f32 rows and with the
operand image.  Time should grow monotonically along both axes; a cell that costs more than a larger neighbour marks a route boundary
set in the wrong place (scan kernel <-> batched kernel <-> image).  python tools/sweep_routes.py > gpurun_out/sweep_routes.json"""
import gc, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import semtools_amd as smt

gc.disable()   # (a full collection with torch imported takes 30-45 ms and lands in the middle of a timing loop)
dev = torch.device("cuda", 0)
ctx = smt.Context(0)
g = torch.Generator(device=dev); g.manual_seed(3)
rows = 30_000_000
x = torch.empty((rows, 256), device=dev)
for b in range(0, rows, 2_000_000):
    c = torch.randn(2_000_000, 256, device=dev, generator=g); c /= c.norm(dim=1, keepdim=True); x[b:b + 2_000_000] = c
del c
g.manual_seed(4)
q = torch.randn(64, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
qh = np.ascontiguousarray(q.cpu().numpy())
sizes = (1000, 10_000, 65_536, 100_000, 300_000, 1_000_000, 1_400_000, 1_600_000, 3_000_000, 10_000_000, 30_000_000)
nqs = (1, 2, 3, 4, 5, 6, 7, 8, 16, 32, 64)
out = {}
import ctypes as C
from semtools_amd import _lib as L
o_rows = np.empty((64, 10), dtype=np.uint64); o_dist = np.empty((64, 10), dtype=np.float64); o_cnt = np.zeros(64, dtype=np.uint64)
for image in (False, True):
    table = {}
    for n in sizes:
        corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=n)
        if image: corpus.prepack()
        row = []
        for nq in nqs:
            def call():
                L.check(L.lib().smt_search(corpus._h, L.np_ptr(qh), nq, 10, float("nan"), smt.MODE_DOCUMENTS, None, 0, 0,
                                           L.np_ptr(o_rows), L.np_ptr(o_dist), L.np_ptr(o_cnt), 10))
            call(); call(); ctx.synchronize()
            reps = 20 if n <= 1_000_000 else 5
            t0 = time.perf_counter()
            for _ in range(reps): call()
            row.append(round((time.perf_counter() - t0) / reps * 1e3, 4))
        table[str(n)] = row
        print(("image " if image else "f32   ") + f"{n:>9}", row, file=sys.stderr)
        corpus.close()
    out["image" if image else "f32_rows"] = {"nq": list(nqs), "ms_by_rows": table}
    # non-monotone cells
    bad = []
    for si, n in enumerate(sizes):
        for qi, nq in enumerate(nqs):
            v = table[str(n)][qi]
            if qi + 1 < len(nqs) and table[str(n)][qi + 1] < 0.9 * v: bad.append(f"rows={n}: nq={nq} {v} ms > nq={nqs[qi + 1]} {table[str(n)][qi + 1]} ms")
            if si + 1 < len(sizes) and table[str(sizes[si + 1])][qi] < 0.9 * v: bad.append(f"nq={nq}: rows={n} {v} ms > rows={sizes[si + 1]} {table[str(sizes[si + 1])][qi]} ms")
    out["image" if image else "f32_rows"]["non_monotone"] = bad
    for b in bad: print("NON-MONOTONE", "image" if image else "f32", b, file=sys.stderr)
print(json.dumps(out, indent=1))
