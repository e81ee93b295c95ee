"""One IVF index over ROWS rows (bench.py's c5 generator), a few 1000-query searches: what tools/trace_ivf.sh traces.
Prints list-length statistics and the wall time of the last search."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt  # noqa: E402
from tests import synth  # noqa: E402

rows = int(os.environ.get("ROWS", 10_000_000))
nq, k, nlist = int(os.environ.get("NQ", 1000)), 10, int(os.environ.get("NLIST", 4096))
nprobe, rerank = int(os.environ.get("NPROBE", 8)), int(os.environ.get("RERANK", 128))
dev = torch.device("cuda:0")
gen = synth.clustered_model_torch(20000, 8, 11, dev)
x = synth.clustered_sample_torch(gen, rows, 12)
q = synth.clustered_sample_torch(gen, nq, 13).cpu().numpy()
del gen
torch.cuda.synchronize()
ctx = smt.Context(0)
whole = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
exact = whole.search(q, top_k=k) if os.environ.get("RECALL", "1") == "1" else None
ix = smt.IvfPq(whole, nlist=nlist, train_iters=10, local_pca=os.environ.get("LPCA", "1") == "1")
sizes = ix.list_sizes().astype(np.int64)
res = {"rows": rows, "nlist": nlist, "nprobe": nprobe, "rerank": rerank, "nq": nq,
       "list_len": {"mean": float(sizes.mean()), "p50": int(np.percentile(sizes, 50)), "p99": int(np.percentile(sizes, 99)), "max": int(sizes.max()),
                    "empty": int((sizes == 0).sum())}}
if os.environ.get("GC_OFF") == "1":      # (is the one slow call of a series Python's cyclic collector?  Not shown: ten calls without the exact
    # batch in front measure 1.6-2.6 ms each with the collector on or off; the 33 ms call only ever appeared right after that batch)
    import gc
    gc.disable()
res["ms_of_every_call"] = []
for _ in range(int(os.environ.get("REPS", 4))):
    t0 = time.perf_counter()
    got = ix.search(q, top_k=k, nprobe=nprobe, rerank=rerank)
    res["ms_per_batch"] = (time.perf_counter() - t0) * 1e3
    res["ms_of_every_call"].append(round(res["ms_per_batch"], 3))
if exact is not None:
    res["recall_at_10"] = sum(len(set(r.tolist()) & set(e.tolist())) for (r, _), (e, _) in zip(got, exact)) / (nq * k)
print(json.dumps(res))
