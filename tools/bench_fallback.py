"""Cost of the exhaustive re-answer in batch mode: nq queries x 10 M rows, every query sitting on a cluster of 20
near-ties around its 10th place (wider than the guard band of 8).  One batched threshold pass (the default) against one
K4 scan per uncertain query (fallback_batch_min_rows set out of reach), same process, same corpus."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt  # noqa: E402


def main():
    rows, nq = 10_000_000, 64
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.empty(rows, 256, device=dev)
    for b in range(0, rows, 2_000_000):
        c = torch.randn(2_000_000, 256, device=dev, generator=g)
        x[b:b + 2_000_000] = c / c.norm(dim=1, keepdim=True)
    q = torch.randn(nq, 256, device=dev, generator=g)
    q /= q.norm(dim=1, keepdim=True)
    spots = torch.randperm(rows, device=dev, generator=g)[: nq * 20].view(nq, 20)
    for i in range(nq):
        v = q[i] + 0.8 * torch.nn.functional.normalize(torch.randn(256, device=dev, generator=g), dim=0)
        v /= v.norm()
        scale = 0.5 + 0.1 * torch.arange(20, device=dev, dtype=torch.float32)
        x[spots[i]] = v[None, :] * scale[:, None]            # rescaled copies: equal cosine up to f32 noise
    torch.cuda.synchronize()
    ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
    qh = q.cpu().numpy()
    out = {}
    for name, min_rows in (("batched_pass", 100000), ("k4_scan_per_query", 1 << 40), ("batched_pass", 100000)):
        ctx.set_tuning("fallback_batch_min_rows", min_rows)
        corpus.search(qh, top_k=10)
        t0 = time.perf_counter()
        for _ in range(3):
            got = corpus.search(qh, top_k=10)
        dt = (time.perf_counter() - t0) / 3
        in_cluster = sum(int(np.isin(got[i][0][:10], spots[i].cpu().numpy()).sum() > 0) for i in range(nq))
        print(json.dumps(dict(mode=name, nq=nq, rows=rows, ms_per_call=round(dt * 1e3, 2), queries_touching_their_cluster=in_cluster)), flush=True)
        out[name] = [g_[0].tolist() for g_ in got]
    print(json.dumps(dict(same_answers=out["batched_pass"] == out["k4_scan_per_query"])))


if __name__ == "__main__":
    main()
