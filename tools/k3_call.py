"""Driver for kernel timelines: `reps` device-resident top-10 calls of nq queries over `rows` random unit rows, with the operand image
(default) or from the f32 rows (--no-image), optionally over a document subset (--subset: every second 1000-row document, host
API).  Prints the wall ms per call.  Run under rocprofv3 --kernel-trace by tools/trace_call.sh."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nq", type=int, default=1000)
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--no-image", action="store_true")
ap.add_argument("--subset", action="store_true")
ap.add_argument("--tune", action="append", default=[])
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(3)
x = torch.empty(a.rows, 256, device=dev)
for b in range(0, a.rows, 2_000_000):
    c = torch.randn(min(2_000_000, a.rows - b), 256, device=dev, generator=g)
    x[b:b + c.shape[0]] = c / c.norm(dim=1, keepdim=True)
q = torch.randn(a.nq, 256, device=dev, generator=g)
q /= q.norm(dim=1, keepdim=True)
torch.cuda.synchronize()
ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
for kv in a.tune:
    key, val = kv.split("=")
    ctx.set_tuning(key, int(val))
corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=a.rows)
if not a.no_image:
    corpus.prepack()
o_r = torch.empty(a.nq, 10, dtype=torch.int64, device=dev)
o_d = torch.empty(a.nq, 10, dtype=torch.float64, device=dev)
if a.subset:
    pr = smt.PackedRanges([(d * 1000, (d + 1) * 1000) for d in range(0, a.rows // 1000, 2)])
    qh = q.cpu().numpy()

    def call():
        corpus.search(qh, top_k=10, ranges=pr)
else:
    def call():
        corpus.search_topk_device(q.data_ptr(), a.nq, 10, 0, o_r.data_ptr(), o_d.data_ptr())
for _ in range(2):
    call()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    call()
ctx.synchronize()
print(json.dumps({"nq": a.nq, "rows": a.rows, "image": not a.no_image, "subset": a.subset, "ms_per_call": (time.perf_counter() - t0) / a.reps * 1e3}))
