"""Phase timing of final_select_kernel (query 0) via s_memtime stamps; run on the GPU box."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn(1_000_000, 256, device=dev, generator=g); x /= x.norm(dim=1, keepdim=True)
q = torch.randn(1, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
for k in (10, 56):
    out_rows = torch.empty(1, k, dtype=torch.int64, device=dev); out_dist = torch.empty(1, k, dtype=torch.float64, device=dev)
    dbg = torch.zeros(16, dtype=torch.int64, device=dev)
    ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=x.shape[0])
    ctx.set_tuning("select_debug_ptr", dbg.data_ptr())
    for _ in range(3):
        corpus.search_topk_device(q.data_ptr(), 1, k, 0, out_rows.data_ptr(), out_dist.data_ptr())
    torch.cuda.synchronize()
    t = dbg.cpu().tolist()
    names = ["load cols", "tau ranks", "compact", "rank surv", "stage rows", "rescore", "final rank"]
    print(f"k={k} survivors={t[8]} total_cycles={t[7]-t[0]} (100MHz ticks? see below)")
    for i, n in enumerate(names):
        print(f"  {n:12s} {t[i+1]-t[i]}")
