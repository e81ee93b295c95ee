"""From how many rows does a corpus that HAS its fp16 operand image answer ONE or TWO queries faster through the batched kernel over
the image (512 B per row + the fixed cost of its levels) than through the scan kernel over the f32 rows (1 KiB per row)?
(search.cpp topk_dispatch, tuning key image_scan_min_rows.)  Device-resident top-10 calls, wall us per call."""
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
for rows in (500_000, 1_000_000, 1_500_000, 2_000_000, 3_000_000, 4_000_000, 6_000_000):
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.randn(rows, 256, device=dev, generator=g)
    x /= x.norm(dim=1, keepdim=True)
    q = torch.randn(8, 256, device=dev, generator=g)
    c = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
    c.prepack()
    o_r = torch.empty(8, 10, dtype=torch.int64, device=dev)
    o_d = torch.empty(8, 10, dtype=torch.float64, device=dev)
    for nq in (1, 2):
        row = {"rows": rows, "nq": nq}
        for name, min_rows in (("scan_f32_us", 0), ("image_us", 1)):
            ctx.set_tuning("image_scan_min_rows", min_rows)
            for _ in range(5):
                c.search_topk_device(q.data_ptr(), nq, 10, 0, o_r.data_ptr(), o_d.data_ptr())
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                c.search_topk_device(q.data_ptr(), nq, 10, 0, o_r.data_ptr(), o_d.data_ptr())
            ctx.synchronize()
            row[name] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
        ctx.set_tuning("image_scan_min_rows", 1_500_000)
        out.append(row)
        print(json.dumps(row), flush=True)
    c.close()
    del x
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_image_scan_sweep.json"), "w"), indent=1)
