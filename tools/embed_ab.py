"""K1 A/B in one process (boxes differ by more than the effect): tuning key embed_batched = 1 (parked lines, the round-3 kernel) and
3 (+ token ids prefetched one step ahead; the default), alternating (the round-4 probes 7 = three waves per SIMD and 11 = nontemporal row
loads were measured with this script -- profiles/r04_k1/ -- and removed from the library);
Zipf ids over a 500 k-row table and uniform ids over a 4 M-row table, 2 M ragged lines -- kernel ms (HIP events) per variant, with the
shader / memory clocks and the power draw rocm-smi reports beside them."""
import json
import subprocess
import sys

import torch

sys.path.insert(0, "/root/repo")
import bench  # noqa: E402
import semtools_amd as smt  # noqa: E402


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        return {k: v for k, v in card.items() if any(s in k.lower() for s in ("sclk", "mclk", "power", "fclk"))}
    except Exception as exc:
        return {"error": repr(exc)}


dev = torch.device("cuda:0")
ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
out = {"idle": smi()}
variants = [int(v) for v in sys.argv[1:]] or [1, 3]
for rnd in range(3):
    for b in variants:
        ctx.set_tuning("embed_batched", b)
        e = bench.bench_embed(smt, ctx, dev, 2_000_000)
        out.setdefault(f"embed_batched={b}", []).append({"zipf_ms": round(e["zipf_ids_500k_table"]["kernel_ms"], 3),
                                                          "uniform_ms": round(e["uniform_ids_4M_table"]["kernel_ms"], 3),
                                                          "uniform_frac_hbm": round(e["uniform_ids_4M_table"]["roofline"]["frac"], 4),
                                                          "bit_exact": e["zipf_ids_500k_table"]["checks"]["sample_lines_bit_exact_vs_oracle"]
                                                          and e["uniform_ids_4M_table"]["checks"]["sample_lines_bit_exact_vs_oracle"]})
    out[f"after_round_{rnd}"] = smi()
print(json.dumps(out, indent=1))
