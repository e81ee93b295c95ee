"""Is the 1000-query K3 batch limited by the MFMA pipe's schedule or by the part's power budget?  Same kernel, same
instruction stream, same candidate statistics, two corpora: unit Gaussian rows, and rows of random signs / 16 (also unit
norm, cosines distributed alike) whose bf16 lo parts are all zero, so one of the three MFMAs of every K-step multiplies
zeros and the hi operands carry one of two bit patterns.  If the batch time follows the data, the clock does.
(Degenerate corpora -- zeros, one repeated row -- are no probe: every row ties, the candidate buffers overflow.)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt  # noqa: E402


def main():
    rows, nq, k = 10_000_000, 1000, 10
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    q = torch.randn(nq, 256, device=dev, generator=g)
    q /= q.norm(dim=1, keepdim=True)
    ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    x = torch.empty(rows, 256, device=dev)
    out_rows = torch.empty(nq, k, dtype=torch.int64, device=dev)
    out_dist = torch.empty(nq, k, dtype=torch.float64, device=dev)
    for name in ("random", "signs", "random", "signs"):
        if name == "random":
            g.manual_seed(3)
            for b in range(0, rows, 2_000_000):
                c = torch.randn(2_000_000, 256, device=dev, generator=g)
                x[b:b + 2_000_000] = c / c.norm(dim=1, keepdim=True)
        else:
            g.manual_seed(7)
            for b in range(0, rows, 2_000_000):
                c = torch.randn(2_000_000, 256, device=dev, generator=g)
                x[b:b + 2_000_000] = torch.sign(c) / 16.0
        torch.cuda.synchronize()
        corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
        ctx.prof_enable(True)
        corpus.search_topk_device(q.data_ptr(), nq, k, 0, out_rows.data_ptr(), out_dist.data_ptr())
        ctx.synchronize()
        ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(5):
            corpus.search_topk_device(q.data_ptr(), nq, k, 0, out_rows.data_ptr(), out_dist.data_ptr())
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / 5
        n_g, ms_g = ctx.prof_read("gemm")
        ctx.prof_enable(False)
        print(json.dumps(dict(corpus=name, wall_ms=round(wall * 1e3, 3), gemm_ms=round(ms_g / 5, 3))), flush=True)
        corpus.close()


if __name__ == "__main__":
    main()
