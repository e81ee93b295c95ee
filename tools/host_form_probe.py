import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import semtools_amd as smt
rng = np.random.default_rng(1)
rows = rng.standard_normal((2_000_000, 256)).astype(np.float32); rows /= np.linalg.norm(rows, axis=1, keepdims=True)
q = rng.standard_normal((1024, 256)).astype(np.float32)
ctx = smt.Context(0)
c = smt.Corpus(ctx); c.append(rows)
for nq in (1, 64, 1024):
    c.search(q[:nq], top_k=3)
    t0 = time.perf_counter()
    for _ in range(5): c.search(q[:nq], top_k=3)
    print(nq, "host-form ms per call", round((time.perf_counter() - t0) / 5 * 1e3, 3))
    ctx.prof_enable(True); ctx.prof_reset(); c.search(q[:nq], top_k=3)
    print("   gemm", ctx.prof_read("gemm"), "select", ctx.prof_read("select"), "scan", ctx.prof_read("scan"))
    ctx.prof_enable(False)
