"""K3 batches (2..1000+ queries x N rows): per batch size the wall time, the summed "gemm" kernel time (HIP events),
the HBM fraction of one corpus pass, the algorithmic flop rate over the f32-MFMA peak, and an all-queries check against
the single-query K2 path on the device.  --tune key=value selects the mode (gemm_bf16x3, gemm_rowreg, ...); runs that
are to be compared belong in ONE gpurun call (boxes differ by more than small deltas)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--nq", type=int, nargs="+", default=[8, 16, 32, 64, 128])
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--variants", type=int, nargs="+", default=[1],
                    help="1 = default routing, 2 = gemm_level_kernel instead of the LDS-row kernel (f32 / gemm_rowreg=0 modes)")
    ap.add_argument("--ranges", action="store_true", help="also time a range-filtered batch (two ranges, 90 %% of the rows)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--live-dims", type=int, default=256, help="rows (and queries) are zero outside their first N dims: the same instruction "
                    "stream over operands that barely switch (power probe)")
    ap.add_argument("--prepack", action="store_true", help="build the corpus' fp16 operand image first (smt_corpus_prepack)")
    ap.add_argument("--tune", action="append", default=[], help="key=value for smt_set_tuning (repeatable)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.empty(args.rows, 256, device=dev)
    for b in range(0, args.rows, 2_000_000):
        c = torch.randn(min(2_000_000, args.rows - b), 256, device=dev, generator=g)
        c[:, args.live_dims:] = 0
        x[b:b + c.shape[0]] = c / c.norm(dim=1, keepdim=True)
    g.manual_seed(5)
    qall = torch.randn(max(args.nq), 256, device=dev, generator=g)
    qall[:, args.live_dims:] = 0
    qall /= qall.norm(dim=1, keepdim=True)
    torch.cuda.synchronize()
    ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=args.rows)
    if args.prepack:
        t0 = time.perf_counter()
        corpus.prepack()
        ctx.synchronize()
        print(json.dumps({"prepack_ms": round((time.perf_counter() - t0) * 1e3, 2), "image_bytes": corpus.image_bytes}))
    for kv in args.tune:
        key, val = kv.split("=")
        ctx.set_tuning(key, int(val))
    results = []
    for nq in args.nq:
        q = qall[:nq].contiguous()
        # truth for every query: the single-query scan path (K2, <= 4 queries per pass)
        k2_rows = torch.empty(nq, args.k, dtype=torch.int64, device=dev)
        k2_dist = torch.empty(nq, args.k, dtype=torch.float64, device=dev)
        ctx.set_tuning("gemm_min_nq", 8)   # (3 .. 7 queries would take K3 too on a shard this size)
        for i in range(0, nq, 4):
            n = min(4, nq - i)
            corpus.search_topk_device(q[i:i + n].data_ptr(), n, args.k, 0, k2_rows[i:i + n].data_ptr(), k2_dist[i:i + n].data_ptr())
        ctx.synchronize()
        ctx.set_tuning("gemm_min_nq", 5)
        for variant in args.variants:
            # 1 = default routing; 2 = (with --tune gemm_rowreg=0 or gemm_bf16x3=0) gemm_level_kernel for every size
            # instead of the LDS-row kernel up to 64 queries.  (Variant 0, the first-generation resident-query
            # kernel, is gone; profiles/r02_k3_small_batch*.json hold its numbers.)
            ctx.set_tuning("gemm_ldsrow", 0 if variant == 2 else 1)
            out_rows = torch.empty(nq, args.k, dtype=torch.int64, device=dev)
            out_dist = torch.empty(nq, args.k, dtype=torch.float64, device=dev)
            ctx.prof_enable(True)
            corpus.search_topk_device(q.data_ptr(), nq, args.k, 0, out_rows.data_ptr(), out_dist.data_ptr())
            ctx.synchronize()
            ctx.prof_reset()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                corpus.search_topk_device(q.data_ptr(), nq, args.k, 0, out_rows.data_ptr(), out_dist.data_ptr())
            ctx.synchronize()
            wall = (time.perf_counter() - t0) / args.reps
            n_g, ms_g = ctx.prof_read("gemm")
            n_s, ms_s = ctx.prof_read("select")
            ctx.prof_enable(False)
            gemm_ms = ms_g / args.reps if n_g else wall * 1e3   # (no K3 launch: the batch went through K2 passes)
            same = int(((k2_rows == out_rows).all(dim=1) & (k2_dist == out_dist).all(dim=1)).sum().item())
            passes = -(-nq // 64) if (variant == 1 and nq <= 128) else 1
            r = dict(rows=args.rows, nq=nq, gemm_ldsrow=variant, wall_ms=round(wall * 1e3, 3), gemm_ms=round(gemm_ms, 3),
                     select_ms=round(ms_s / args.reps, 3), gemm_launches=n_g // args.reps, corpus_passes=passes,
                     hbm_frac_one_pass=round(args.rows * 1024 / (gemm_ms * 1e-3) / 8e12, 3),
                     mfma_frac=round(2.0 * nq * args.rows * 256 / (gemm_ms * 1e-3) / 157.3e12, 3),
                     qps=round(nq / wall, 1), k2_agreement=f"{same}/{nq}")
            results.append(r)
            print(json.dumps(r), flush=True)
        if args.ranges:
            ctx.set_tuning("gemm_ldsrow", 1)
            cut = args.rows // 20
            rng = [(cut, args.rows // 2), (args.rows // 2 + cut, args.rows)]
            qh = q.cpu().numpy()
            corpus.search(qh, top_k=args.k, ranges=rng)
            t0 = time.perf_counter()
            got = corpus.search(qh, top_k=args.k, ranges=rng)
            dt = time.perf_counter() - t0
            one = [corpus.search(qh[i], top_k=args.k, ranges=rng)[0] for i in range(min(nq, 8))]
            ok = all(got[i][0].tolist() == one[i][0].tolist() for i in range(len(one)))
            r = dict(rows=args.rows, nq=nq, ranged_rows=rng[0][1] - rng[0][0] + rng[1][1] - rng[1][0],
                     host_call_ms=round(dt * 1e3, 3), matches_single_query_path=ok)
            results.append(r)
            print(json.dumps(r), flush=True)
    if args.out:
        json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
