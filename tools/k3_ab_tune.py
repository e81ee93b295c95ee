"""A/B of one tuning key on batched searches in ONE process (same corpus, alternating): ms per device-resident call for each value of
the key, several batch sizes, with the operand image and from the f32 rows.
    python tools/k3_ab_tune.py gemm_split_last 1 2 [--rows 10000000] [--nq 1000 256]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("key")
ap.add_argument("values", type=int, nargs="+")
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--nq", type=int, nargs="+", default=[1000, 256])
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(3)
x = torch.empty(a.rows, 256, device=dev)
for b in range(0, a.rows, 2_000_000):
    c = torch.randn(min(2_000_000, a.rows - b), 256, device=dev, generator=g)
    x[b:b + c.shape[0]] = c / c.norm(dim=1, keepdim=True)
torch.cuda.synchronize()
ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
out = {}
for image in (True, False):
    corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=a.rows)
    if image:
        corpus.prepack()
    for nq in a.nq:
        q = torch.randn(nq, 256, device=dev, generator=g)
        q /= q.norm(dim=1, keepdim=True)
        o_r = torch.empty(nq, 10, dtype=torch.int64, device=dev)
        o_d = torch.empty(nq, 10, dtype=torch.float64, device=dev)
        ref = None
        for rnd in range(3):
            for v in a.values:
                ctx.set_tuning(a.key, v)
                for _ in range(2):
                    corpus.search_topk_device(q.data_ptr(), nq, 10, 0, o_r.data_ptr(), o_d.data_ptr())
                ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.reps):
                    corpus.search_topk_device(q.data_ptr(), nq, 10, 0, o_r.data_ptr(), o_d.data_ptr())
                ctx.synchronize()
                ms = (time.perf_counter() - t0) / a.reps * 1e3
                out.setdefault(f"{'image' if image else 'f32rows'} nq={nq} {a.key}={v}", []).append(round(ms, 4))
                if ref is None:
                    ref = o_r.clone()
                else:
                    assert bool((ref == o_r).all().item()), "answers differ between the variants"
    corpus.close()
print(json.dumps(out, indent=1))
