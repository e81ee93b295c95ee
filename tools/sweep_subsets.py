"""Workspace-mode searches over document subsets of a 10 M-row corpus: document length x wanted fraction x query count, with the operand
image; ms per host call and the cost per scanned row relative to the unfiltered call of the same query count.  Looks for cliffs at the
borders between the kernels (tile table / chunk table / scan kernel).  python tools/sweep_subsets.py > gpurun_out/sweep_subsets.json"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import semtools_amd as smt

dev = torch.device("cuda", 0)
ctx = smt.Context(0)
g = torch.Generator(device=dev); g.manual_seed(3)
rows = 10_000_000
x = torch.empty((rows, 256), device=dev)
for b in range(0, rows, 2_000_000):
    c = torch.randn(2_000_000, 256, device=dev, generator=g); c /= c.norm(dim=1, keepdim=True); x[b:b + 2_000_000] = c
del c
g.manual_seed(6)
q = torch.randn(256, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True); q = np.ascontiguousarray(q.cpu().numpy())
corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
corpus.prepack()
rng = np.random.default_rng(1)

def timed(nq, ranges):
    kw = dict(top_k=10, max_distance=0.9, mode=smt.MODE_WORKSPACE, ranges=ranges)
    corpus.search(q[:nq], **kw); corpus.search(q[:nq], **kw)
    ctx.synchronize()
    ctx.uncertain_count()
    t0 = time.perf_counter()
    for _ in range(5): got = corpus.search(q[:nq], **kw)
    return (time.perf_counter() - t0) / 5 * 1e3, int(ctx.uncertain_count()), got

out = {}
base = {nq: timed(nq, None)[0] for nq in (1, 16, 256)}
out["unfiltered_ms"] = base
for doc_len in (4, 40, 1000):
    n_docs = rows // doc_len
    for frac in (0.01, 0.05, 0.25, 0.5, 0.9):
        want = np.sort(rng.choice(n_docs, size=max(1, int(n_docs * frac)), replace=False))
        ranges = smt.PackedRanges([(int(d) * doc_len, (int(d) + 1) * doc_len) for d in want])
        scanned = len(want) * doc_len
        for nq in (1, 16, 256):
            ms, unc, got = timed(nq, ranges)
            # check the first query's best hit against fp64 over the eligible rows (chunks)
            key = f"doc_len={doc_len} frac={frac} nq={nq}"
            out[key] = {"ms": round(ms, 3), "n_ranges": len(want), "scanned_rows": scanned, "uncertain": unc,
                        "cost_per_scanned_row_vs_unfiltered": round((ms / scanned) / (base[nq] / rows), 2)}
            print(key, out[key], file=sys.stderr)
print(json.dumps(out, indent=1))
