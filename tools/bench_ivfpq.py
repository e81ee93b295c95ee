"""Config c5 on ONE GPU: IVF-PQ (nlist=4096, m=32) over a two-level clustered corpus generated on the device.
Reports build time (per phase), index size, and for several nprobe: recall@10 vs the exact search, queries/s."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt
from tests import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nq", type=int, default=1000)
    ap.add_argument("--k", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    x = synth.clustered_rows_torch(args.rows, 4096, 8, 11, dev)
    g = torch.Generator(device=dev); g.manual_seed(12)
    qi = torch.randint(0, args.rows, (args.nq,), device=dev, generator=g)
    q = x[qi] + 0.002 * torch.randn(args.nq, 256, device=dev, generator=g)
    torch.cuda.synchronize()
    ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=args.rows)
    qh = q.cpu().numpy()
    t0 = time.perf_counter()
    ix = smt.IvfPq(corpus, nlist=args.nlist, train_iters=10)
    build_s = time.perf_counter() - t0
    info = ix.info()
    sizes = ix.list_sizes()
    t0 = time.perf_counter(); exact = corpus.search(qh, top_k=args.k); exact_s = time.perf_counter() - t0
    print(json.dumps(dict(rows=args.rows, nlist=args.nlist, build_s=round(build_s, 3), build_ms=info["build_ms"],
                          index_MB=round(info["index_bytes"] / 1e6, 1), list_size_mean=float(sizes.mean()), list_size_max=int(sizes.max()),
                          exact_batch_s=round(exact_s, 4), exact_qps=round(args.nq / exact_s, 1))), flush=True)
    for nprobe, rerank in ((1, 0), (4, 0), (8, 0), (32, 0), (8, 64), (8, 512), (32, 512), (128, 512)):
        ix.search(qh[:8], top_k=args.k, nprobe=nprobe, rerank=rerank)
        ctx.prof_enable(True); ctx.prof_reset()
        t0 = time.perf_counter()
        got = ix.search(qh, top_k=args.k, nprobe=nprobe, rerank=rerank)
        dt = time.perf_counter() - t0
        n_a, ms_a = ctx.prof_read("ivf_adc"); n_p, ms_p = ctx.prof_read("ivf_probe"); n_s, ms_s = ctx.prof_read("select")
        ctx.prof_enable(False)
        hit = sum(len(set(r.tolist()) & set(e.tolist())) for (r, _), (e, _) in zip(got, exact))
        scanned = float(sizes.mean()) * nprobe * args.nq
        print(json.dumps(dict(nprobe=nprobe, rerank=rerank or "default", recall_at_k=round(hit / (args.nq * args.k), 4),
                              batch_ms=round(dt * 1e3, 3), qps=round(args.nq / dt, 1), probe_ms=round(ms_p, 3), adc_ms=round(ms_a, 3),
                              select_ms=round(ms_s, 3), adc_codes_GBps=round(scanned * 32 / (ms_a * 1e-3) / 1e9, 1) if ms_a else None)), flush=True)


if __name__ == "__main__":
    main()
