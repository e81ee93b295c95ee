"""A/B of the graded bootstrap stride (gemm_topk.hip: every 2nd / 4th / 8th tile for corpora of 2 Ki .. 32 Ki tiles) against every 16th:
ms per host-form top-10 call, 16 / 64 / 256 / 1000 queries, answers compared."""
import gc, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import semtools_amd as smt
from semtools_amd import _lib as L
gc.disable()
dev = torch.device("cuda", 0)
ctx = smt.Context(0)
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn(2_000_000, 256, device=dev, generator=g); x /= x.norm(dim=1, keepdim=True)
g.manual_seed(5)
q = torch.randn(1000, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
qh = np.ascontiguousarray(q.cpu().numpy())
o_rows = np.empty((1000, 10), dtype=np.uint64); o_dist = np.empty((1000, 10), dtype=np.float64); o_cnt = np.zeros(1000, dtype=np.uint64)
ref = {}
for image in (False, True):
  for n in (66_000, 100_000, 131_072, 200_000, 262_144, 400_000, 524_288, 1_000_000):
    corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=n)
    if image: corpus.prepack()
    res = {}
    for stride in ("every 16th (gemm_boot_fine=0)", "graded (default)"):
        ctx.set_tuning("gemm_boot_fine", 0 if stride.startswith("every") else 1)
        row = []
        for nq in (16, 64, 256, 1000):
            def call():
                L.check(L.lib().smt_search(corpus._h, L.np_ptr(qh), nq, 10, float("nan"), smt.MODE_DOCUMENTS, None, 0, 0, L.np_ptr(o_rows), L.np_ptr(o_dist), L.np_ptr(o_cnt), 10))
            call(); call(); ctx.uncertain_count()
            key = (n, nq)
            if key not in ref: ref[key] = (o_rows[:nq].copy(), o_dist[:nq].copy())
            same = np.array_equal(ref[key][0], o_rows[:nq]) and np.array_equal(ref[key][1], o_dist[:nq])
            t0 = time.perf_counter()
            for _ in range(5): call()
            row.append(round((time.perf_counter() - t0) / 5 * 1e3, 3) if same else "DIFF")
        res[stride] = row
    print(f"image={int(image)} rows={n:>8}", res, "uncertain", ctx.uncertain_count())
    corpus.close()
