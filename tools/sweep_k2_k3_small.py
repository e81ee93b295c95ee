"""Where does the batched kernel (K3) overtake the scan kernel (K2) for 2..7 queries?  (search.cpp topk_dispatch: tuning keys
gemm_min_nq / gemm_min_rows_small.)  Device-resident top-10 calls on random unit rows, no operand image (an adopted corpus), wall us
per call for both kernels at each (rows, queries)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt  # noqa: E402

dev = torch.device("cuda:0")
ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
out = []
for rows in (100_000, 200_000, 400_000, 700_000, 1_000_000, 2_000_000):
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.randn(rows, 256, device=dev, generator=g)
    x /= x.norm(dim=1, keepdim=True)
    q = torch.randn(8, 256, device=dev, generator=g)
    c = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
    o_r = torch.empty(8, 10, dtype=torch.int64, device=dev)
    o_d = torch.empty(8, 10, dtype=torch.float64, device=dev)
    for nq in (1, 2, 3, 4, 5, 7):
        row = {"rows": rows, "nq": nq}
        for name, keys in (("k2", {"gemm_min_nq": 8}), ("k3", {"gemm_min_nq": 2, "gemm_min_rows_small": 1})):
            if nq == 1 and name == "k3":
                continue
            for kk, vv in keys.items():
                ctx.set_tuning(kk, vv)
            for _ in range(5):
                c.search_topk_device(q.data_ptr(), nq, 10, 0, o_r.data_ptr(), o_d.data_ptr())
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                c.search_topk_device(q.data_ptr(), nq, 10, 0, o_r.data_ptr(), o_d.data_ptr())
            ctx.synchronize()
            row[name + "_us"] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
            ctx.set_tuning("gemm_min_nq", 5)
            ctx.set_tuning("gemm_min_rows_small", 1_000_000)
        out.append(row)
        print(json.dumps(row), flush=True)
    c.close()
    del x
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_k2_k3_small.json"), "w"), indent=1)
