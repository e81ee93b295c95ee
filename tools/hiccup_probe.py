"""A one-off 37 ms call was seen as the SECOND 1024-query call after a series of 64-query calls: which kernels run in it?"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import semtools_amd as smt
rng = np.random.default_rng(1)
base = rng.standard_normal((50_000, 256)).astype(np.float32)
q = rng.standard_normal((1024, 256)).astype(np.float32)
ctx = smt.Context(0)
rows = rng.standard_normal((2_000_000, 256)).astype(np.float32); rows /= np.linalg.norm(rows, axis=1, keepdims=True)
c = smt.Corpus(ctx); c.append(rows)
ctx.prof_enable(True)
for nq in (64, 64, 64, 1024, 1024, 1024, 1024):
    ctx.prof_reset()
    t0 = time.perf_counter(); r = c.search(q[:nq], top_k=3); ms = (time.perf_counter() - t0) * 1e3
    print(nq, round(ms, 2), {k: ctx.prof_read(k) for k in ("gemm", "gemm_thr", "select", "scan", "pack_image")}, "uncertain", ctx.uncertain_count())
