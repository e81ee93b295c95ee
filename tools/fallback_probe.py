"""Corpora of repeated lines: 2 M rows = 50 k distinct rows x 40 copies (every answer a 40-way exact tie, wider than any guard band):
the batch's certificate fails for every query and the exhaustive re-answer (search.cpp batched_fallback) takes over.  Wall time per
call and the kernel-time split, beside the same batch over 2 M distinct rows."""
import sys, time, json
import numpy as np
sys.path.insert(0, "/root/repo")
import semtools_amd as smt
rng = np.random.default_rng(1)
base = rng.standard_normal((50_000, 256)).astype(np.float32); base /= np.linalg.norm(base, axis=1, keepdims=True)
q = rng.standard_normal((1024, 256)).astype(np.float32)
ctx = smt.Context(0)
out = {}
for name, rows in (("distinct", None), ("repeated_40x", np.tile(base, (40, 1)))):
    if rows is None:
        rows = rng.standard_normal((2_000_000, 256)).astype(np.float32); rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    c = smt.Corpus(ctx); c.append(rows)
    res = {}
    for nq in (1, 64, 1024):
        c.search(q[:nq], top_k=3)
        t0 = time.perf_counter()
        for _ in range(3): r = c.search(q[:nq], top_k=3)
        ms = (time.perf_counter() - t0) / 3 * 1e3
        ctx.prof_enable(True); ctx.prof_reset(); c.search(q[:nq], top_k=3)
        res[nq] = {"ms_per_call": round(ms, 3), "gemm": ctx.prof_read("gemm"), "gemm_thr": ctx.prof_read("gemm_thr"), "select": ctx.prof_read("select"), "scan": ctx.prof_read("scan"),
                   "first_rows": [int(x) for x in r[0][0]]}
        ctx.prof_enable(False)
    out[name] = res
    c.close()
print(json.dumps(out))
