"""Per-block finish times of a K2 launch INSIDE the async pipeline (select of query i on the aux stream while query i+1 scans): how
far behind the median block is the last one?  (wall_clock64 stamps of the pipeline's last launch, 100 MHz; tools/scan_balance.py is
the same for an isolated launch.)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt

rows = 1_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(3)
x = torch.randn(rows, 256, device=dev, generator=g)
x /= x.norm(dim=1, keepdim=True)
q = torch.randn(16, 256, device=dev, generator=g)
torch.cuda.set_stream(torch.cuda.Stream(dev))
ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
for kv in sys.argv[1:]:   # tuning keys to set first, e.g. scan_steal=0
    ctx.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
blocks, waves = 256, 8
stamps = torch.zeros(2 * blocks * waves + blocks, dtype=torch.int64, device=dev)
out = torch.empty((64, 2, 10), dtype=torch.int64).pin_memory()
pct = lambda a: [round(float(np.percentile(a, p)), 1) for p in (0, 10, 50, 90, 99, 100)]
for mode in (0, 1):
    ctx.set_tuning("async_select", mode)
    ctx.set_tuning("scan_debug_ptr", 0)
    for i in range(600):
        corpus.search_topk_device(q[i % 16].data_ptr(), 1, 10, 0, out[i % 64, 0].data_ptr(), out[i % 64, 1].data_ptr())
    ctx.synchronize()
    res = []
    for rep in range(6):
        ctx.set_tuning("scan_debug_ptr", stamps.data_ptr())
        for i in range(40 + rep):
            corpus.search_topk_device(q[i % 16].data_ptr(), 1, 10, 0, out[i % 64, 0].data_ptr(), out[i % 64, 1].data_ptr())
        ctx.synchronize()
        s = stamps.cpu().numpy()
        w = s[: 2 * blocks * waves].reshape(blocks * waves, 2).astype(np.float64) / 100.0
        bend = s[2 * blocks * waves:].astype(np.float64) / 100.0
        t0 = w[:, 0].min()
        be = bend - t0
        le = (w[:, 1] - t0).reshape(blocks, waves).max(axis=1)
        res.append((float(be.max()), float(np.median(be)), float(be.max() - np.median(be))))
        if rep == 5:
            print(f"async_select={mode}: block end {pct(be)}  block loop-end {pct(le)}  wave start {pct(w[:, 0] - t0)}")
            late = np.argsort(be)[-5:]
            print("   latest blocks (index, end us, XCD):", [(int(b), round(float(be[b]), 1), int(b % 8)) for b in late])
    print(f"async_select={mode}: (last block end, median block end, difference) over 6 launches:", [tuple(round(v, 1) for v in r) for r in res])
