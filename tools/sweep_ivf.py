"""IVF index quality sweep on a corpus whose structure does NOT match the index: n_centers topics (default 20 000)
against nlist = 4096 lists, queries drawn independently from the generative model (not perturbed corpus rows).
For both index kinds (0 = global residual PQ, 1 = per-list PCA codes): build time, then recall@k vs the exact batched
search and device-resident queries/s over nprobe x re-score depth."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt  # noqa: E402
from tests import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--topics", type=int, default=20000)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nq", type=int, default=1000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--kinds", type=int, nargs="+", default=[1, 0])
    ap.add_argument("--nprobe", type=int, nargs="+", default=[1, 2, 4, 8, 16, 32, 64])
    ap.add_argument("--rerank", type=int, nargs="+", default=[32, 64, 128, 512])
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    model = synth.clustered_model_torch(args.topics, 8, 11, dev)
    x = synth.clustered_sample_torch(model, args.rows, 12)
    q = synth.clustered_sample_torch(model, args.nq, 13)
    torch.cuda.synchronize()
    ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=args.rows)
    qh = q.cpu().numpy()
    t0 = time.perf_counter()
    exact = corpus.search(qh, top_k=args.k)
    exact_s = time.perf_counter() - t0
    out = dict(rows=args.rows, topics=args.topics, nlist=args.nlist, nq=args.nq, k=args.k, exact_batch_s=round(exact_s, 4),
               queries="independent draws from the generative model", kinds={})
    o_rows = torch.empty((args.nq, args.k), dtype=torch.int64, device=dev)
    o_dist = torch.empty((args.nq, args.k), dtype=torch.float64, device=dev)
    for kind in args.kinds:
        t0 = time.perf_counter()
        ix = smt.IvfPq(corpus, nlist=args.nlist, train_iters=10, local_pca=bool(kind))
        build_s = time.perf_counter() - t0
        info = ix.info()
        rec = dict(kind="per-list PCA codes" if kind else "global residual PQ", build_s=round(build_s, 3), build_ms=info["build_ms"],
                   index_MB=round(info["index_bytes"] / 1e6, 1), sweep=[])
        print(json.dumps({k: v for k, v in rec.items() if k != "sweep"}), flush=True)
        for nprobe in args.nprobe:
            for rerank in args.rerank:
                ix.search_device(q.data_ptr(), args.nq, args.k, nprobe, rerank, 0, o_rows.data_ptr(), o_dist.data_ptr())
                ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    ix.search_device(q.data_ptr(), args.nq, args.k, nprobe, rerank, 0, o_rows.data_ptr(), o_dist.data_ptr())
                ctx.synchronize()
                dt = (time.perf_counter() - t0) / 3
                got = o_rows.cpu().numpy().view(np.uint64)
                hit = sum(len(set(got[i].tolist()) & set(exact[i][0].tolist())) for i in range(args.nq))
                row = dict(nprobe=nprobe, rerank=rerank, recall=round(hit / (args.nq * args.k), 4), ms_per_batch=round(dt * 1e3, 3),
                           qps=round(args.nq / dt))
                rec["sweep"].append(row)
                print(json.dumps(row), flush=True)
        out["kinds"][str(kind)] = rec
        ix.close()
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
