"""Tuning sweep of the K2 scan on a resident synthetic corpus (run on the GPU box).
Prints one line per configuration: kernel µs (HIP events on the library stream), GB/s."""
import argparse
import itertools
import json
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--full", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.randn(args.rows, 256, device=dev, generator=g)
    x /= x.norm(dim=1, keepdim=True)
    q = torch.randn(4, 256, device=dev, generator=g)
    q /= q.norm(dim=1, keepdim=True)
    out_rows = torch.empty(4, args.k, dtype=torch.int64, device=dev)
    out_dist = torch.empty(4, args.k, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ctx = smt.Context(0)
    corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=args.rows)
    ctx.prof_enable(True)
    bytes_per = args.rows * 1024

    def run(label, nq=1, **tune):
        for k_, v_ in tune.items():
            ctx.set_tuning(k_, v_)
        for _ in range(3):
            corpus.search_topk_device(q.data_ptr(), nq, args.k, 0, out_rows.data_ptr(), out_dist.data_ptr())
        ctx.synchronize()
        ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            corpus.search_topk_device(q.data_ptr(), nq, args.k, 0, out_rows.data_ptr(), out_dist.data_ptr())
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / args.reps
        n, ms = ctx.prof_read("scan")
        n2, ms2 = ctx.prof_read("select")
        us = ms / max(n, 1) * 1e3
        print(json.dumps(dict(label=label, nq=nq, **tune, scan_us=round(us, 2), select_us=round(ms2 / max(n2, 1) * 1e3, 2),
                              wall_us=round(wall * 1e6, 2), scan_GBps=round(bytes_per / (us * 1e-6) / 1e9, 1),
                              wall_GBps=round(bytes_per / wall / 1e9, 1))), flush=True)

    run("default")
    # sanity vs torch fp64
    ref = 1.0 - (x.double() @ q[0].double())
    tv, ti = torch.topk(ref, args.k, largest=False)
    print("rows match torch:", (out_rows[0].cpu() == ti.cpu()).all().item() if True else None,
          "max|dd|", (out_dist[0] - tv).abs().max().item(), flush=True)
    threads_blocks = [(256, 8), (512, 2), (512, 4), (1024, 1), (1024, 2), (256, 4), (512, 1), (256, 16)]
    if not args.full:
        threads_blocks = [(512, 2), (1024, 1), (256, 2), (512, 1)]
    for (threads, bpc), unroll, nt in itertools.product(threads_blocks, (4, 8) if not args.full else (2, 4, 8), (1,) if not args.full else (1, 0)):
        run("sweep", scan_threads=threads, scan_blocks=bpc * 256, scan_unroll=unroll, scan_nontemporal=nt)
    ctx.set_tuning("scan_threads", 1024); ctx.set_tuning("scan_blocks", 0); ctx.set_tuning("scan_unroll", 4)
    ctx.set_tuning("scan_nontemporal", 1)
    for threads, blocks, unroll, pf in ((1024, 256, 4, 1), (1024, 256, 2, 1), (1024, 256, 8, 1), (512, 256, 4, 1), (512, 256, 8, 1),
                                        (512, 512, 4, 1), (512, 512, 2, 1), (256, 512, 8, 1), (1024, 256, 2, 0), (1024, 256, 4, 0)):
        run("prefetch" if pf else "plain", scan_threads=threads, scan_blocks=blocks, scan_unroll=unroll, scan_prefetch=pf)
    ctx.set_tuning("scan_threads", 1024); ctx.set_tuning("scan_blocks", 0); ctx.set_tuning("scan_unroll", 4)
    ctx.set_tuning("scan_prefetch", 0)
    for nq in (2, 4):
        run("multiq", nq=nq)
    # torch reference bandwidth: a plain read-reduce of the same matrix
    torch.cuda.synchronize()
    for fn, name in ((lambda: x.sum(), "torch.sum"), (lambda: x @ q[0], "torch.mv")):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        print(json.dumps(dict(label=name, wall_us=round(dt * 1e6, 2), GBps=round(bytes_per / dt / 1e9, 1))), flush=True)


if __name__ == "__main__":
    main()
