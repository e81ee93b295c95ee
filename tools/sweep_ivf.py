"""The IVF index across its query parameters on a 10 M-row clustered corpus (per-list PCA codes): recall@k against the exact batched
search and ms per device-form call over nprobe x rerank, top_k, and query counts 1 .. 10 000.  Recall must not fall when nprobe or
rerank grow; time must not fall when they grow.  python tools/sweep_ivf.py > gpurun_out/sweep_ivf.json"""
import gc, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import semtools_amd as smt
from tests import synth

gc.disable()
dev = torch.device("cuda", 0)
ctx = smt.Context(0)
rows = 10_000_000
gen = synth.clustered_model_torch(20000, 8, 11, dev)
x = synth.clustered_sample_torch(gen, rows, 12)
q_all = synth.clustered_sample_torch(gen, 10000, 13)
del gen
torch.cuda.synchronize()   # (the library works on its own stream: the rows must exist before the index is built from them)
corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
ix = smt.IvfPq(corpus, nlist=4096, train_iters=10, local_pca=True)
out = {}

def run(nq, k, nprobe, rerank, exact=None):
    qd = q_all[:nq].contiguous()
    o_rows = torch.empty((nq, k), dtype=torch.int64, device=dev); o_dist = torch.empty((nq, k), dtype=torch.float64, device=dev)
    ix.search_device(qd.data_ptr(), nq, k, nprobe, rerank, 0, o_rows.data_ptr(), o_dist.data_ptr()); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): ix.search_device(qd.data_ptr(), nq, k, nprobe, rerank, 0, o_rows.data_ptr(), o_dist.data_ptr())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    rec = None
    if exact is not None:
        got = o_rows.cpu().numpy().view(np.uint64)
        rec = sum(len(set(got[i].tolist()) & set(exact[i][0].tolist())) for i in range(nq)) / (nq * k)
    return round(ms, 3), (round(rec, 4) if rec is not None else None)

qh = q_all[:1000].cpu().numpy()
exact10 = corpus.search(qh, top_k=10)
for nprobe in (1, 4, 8, 32, 128, 512):
    for rerank in (16, 64, 128, 512):
        out[f"nq=1000 k=10 nprobe={nprobe} rerank={rerank}"] = dict(zip(("ms", "recall"), run(1000, 10, nprobe, rerank, exact10)))
        print(f"nprobe={nprobe} rerank={rerank}", out[f"nq=1000 k=10 nprobe={nprobe} rerank={rerank}"], file=sys.stderr)
for k in (1, 3, 10, 30, 56):
    ex = corpus.search(qh[:200], top_k=k)
    out[f"nq=200 k={k} nprobe=8 rerank=128"] = dict(zip(("ms", "recall"), run(200, k, 8, 128, ex)))
    print(f"k={k}", out[f"nq=200 k={k} nprobe=8 rerank=128"], file=sys.stderr)
for nq in (1, 8, 64, 512, 4096, 10000):
    ms, _ = run(nq, 10, 8, 128)
    out[f"nq={nq} k=10 nprobe=8 rerank=128"] = {"ms": ms, "us_per_query": round(ms / nq * 1e3, 2)}
    print(f"nq={nq}", out[f"nq={nq} k=10 nprobe=8 rerank=128"], file=sys.stderr)
print(json.dumps(out, indent=1))
