"""Per-wave / per-block timing of one K2 launch (wall_clock64 stamps, 100 MHz): how uneven is the finish?"""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.environ.get("SMT_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn(rows, 256, device=dev, generator=g); x /= x.norm(dim=1, keepdim=True)
q = torch.randn(256, device=dev, generator=g); q /= q.norm()
ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
blocks, waves = 256, 8
stamps = torch.zeros(2 * blocks * waves + blocks, dtype=torch.int64, device=dev)
out_r = torch.empty(10, dtype=torch.int64, device=dev); out_d = torch.empty(10, dtype=torch.float64, device=dev)
for _ in range(5):
    corpus.search_topk_device(q.data_ptr(), 1, 10, 0, out_r.data_ptr(), out_d.data_ptr())
torch.cuda.synchronize()
ctx.set_tuning("scan_debug_ptr", stamps.data_ptr())
for rep in range(3):
    corpus.search_topk_device(q.data_ptr(), 1, 10, 0, out_r.data_ptr(), out_d.data_ptr())
    torch.cuda.synchronize()
    s = stamps.cpu().numpy()
    w = s[: 2 * blocks * waves].reshape(blocks * waves, 2).astype(np.float64) / 100.0   # us
    bend = s[2 * blocks * waves:].astype(np.float64) / 100.0
    t0 = w[:, 0].min()
    st, le, be = w[:, 0] - t0, w[:, 1] - t0, bend - t0
    pct = lambda a: [round(float(np.percentile(a, p)), 1) for p in (0, 10, 50, 90, 100)]
    print(f"rep {rep}: wave start {pct(st)}  wave loop-end {pct(le)}  block end {pct(be)}  (us; percentiles 0/10/50/90/100)")
    dur = le - st
    print(f"        wave loop duration {pct(dur)}  block tail (end - last wave loop-end) {pct(be - le.reshape(blocks, waves).max(axis=1))}")
    lb = le.reshape(blocks, waves)
    print("        loop-end by wave slot (mean over blocks):", [round(float(v), 1) for v in lb.mean(axis=0)])
    print("        block loop-end (max over its waves) by XCD = block % 8:", [round(float(lb.max(axis=1)[x::8].mean()), 1) for x in range(8)])
    print("        spread inside a block (max-min over waves): ", pct(lb.max(axis=1) - lb.min(axis=1)), " spread of block means:", pct(lb.mean(axis=1)))
ctx.set_tuning("scan_debug_ptr", 0)
