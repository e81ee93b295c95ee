import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import semtools_amd as smt
rng = np.random.default_rng(1)
q = rng.standard_normal((1024, 256)).astype(np.float32)
rows = rng.standard_normal((2_000_000, 256)).astype(np.float32); rows /= np.linalg.norm(rows, axis=1, keepdims=True)
for img in (1, 0):
    ctx = smt.Context(0)
    ctx.set_tuning("corpus_image", img)
    c = smt.Corpus(ctx); c.append(rows)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); c.search(q, top_k=3); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    print("corpus_image", img, ts)
    c.close(); ctx.close() if hasattr(ctx, "close") else None
