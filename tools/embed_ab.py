"""K1 A/B in one process: tuning key embed_batched 1 / 0 alternating, Zipf ids over a 500 k-row table and uniform ids over a 4 M-row
table, 2 M ragged lines -- kernel ms (HIP events) per variant."""
import json, sys
import torch
sys.path.insert(0, "/root/repo")
import bench
import semtools_amd as smt
dev = torch.device("cuda:0")
ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
out = {}
for rnd in range(3):
    for b in (1, 0):
        ctx.set_tuning("embed_batched", b)
        e = bench.bench_embed(smt, ctx, dev, 2_000_000)
        out.setdefault(f"batched={b}", []).append((round(e["zipf_ids_500k_table"]["kernel_ms"], 3), round(e["uniform_ids_4M_table"]["kernel_ms"], 3)))
print(json.dumps(out))
