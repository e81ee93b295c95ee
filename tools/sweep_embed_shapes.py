"""K1 (smt_embed_device) across line-length distributions, 32 M tokens per launch over a 500 k-row table, Zipf and uniform ids: ms,
lines/s, tokens/s, algorithmic GB/s ((tokens + lines) x 1 KiB).  Looks for shapes the run-walking groups handle badly (one-token
lines, 2048-token lines, empty lines, one long line among short ones).  A sample of lines is checked bit for bit against the oracle.
python tools/sweep_embed_shapes.py > gpurun_out/sweep_embed_shapes.json"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import semtools_amd as smt
from oracle import oracle as orc

dev = torch.device("cuda", 0)
ctx = smt.Context(0)
V = 500_000
table = torch.randn(V, 256, device=dev) * 0.1
table_h = table.cpu().numpy()
model = smt.Model(ctx, device_ptr=table.data_ptr(), V=V, normalize=True)
if os.environ.get("EMBED_BATCHED"): ctx.set_tuning("embed_batched", int(os.environ["EMBED_BATCHED"]))   # 11 = runs of equal line counts (A/B)
rng = np.random.default_rng(5)
T = 32_000_000

def lengths(kind):
    if kind.startswith("const"):
        L = int(kind[5:]); return np.full(T // L, L, dtype=np.int64)
    if kind == "ragged_0_32": return rng.integers(0, 33, size=T // 16)
    if kind == "half_empty": x = rng.integers(1, 33, size=T // 8); x[::2] = 0; return x
    if kind == "mostly_4_some_2048":
        x = np.full(T // 8, 4, dtype=np.int64); x[rng.choice(x.size, size=x.size // 500, replace=False)] = 2048; return x
    if kind == "lognormal_median7":       # heavy-tailed like lines of text and code: median 7 tokens, mean ~12, 1 % beyond 75
        return np.clip(np.round(np.exp(rng.normal(2.0, 1.0, size=T // 12))), 0, 4096).astype(np.int64)
    if kind == "one_giant_among_short":
        x = np.full(T // 16, 8, dtype=np.int64); x[x.size // 2] = 4_000_000; return x       # truncated to 2048 by max_tokens
    raise ValueError(kind)

out = {}
SHAPES = os.environ.get("SHAPES", "const1,const4,const16,const64,const512,const2048,ragged_0_32,half_empty,lognormal_median7,mostly_4_some_2048,one_giant_among_short").split(",")
for kind in SHAPES:
    ln = lengths(kind).astype(np.int64)
    offsets = np.zeros(ln.size + 1, dtype=np.int64); np.cumsum(ln, out=offsets[1:])
    n_tok = int(offsets[-1]); n_lines = int(ln.size)
    used = int(np.minimum(ln, 2048).sum())
    for dist in ("zipf", "uniform"):
        ids = ((rng.zipf(1.1, size=n_tok) - 1) % V if dist == "zipf" else rng.integers(0, V, size=n_tok)).astype(np.int32)
        d_ids = torch.from_numpy(ids).to(dev); d_off = torch.from_numpy(offsets).to(dev)
        o = torch.empty((n_lines, 256), device=dev)
        model.embed_device(d_ids.data_ptr(), d_off.data_ptr(), n_lines, 2048, o.data_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ctx.synchronize(); e0.record()
        for _ in range(3): model.embed_device(d_ids.data_ptr(), d_off.data_ptr(), n_lines, 2048, o.data_ptr())
        ctx.synchronize(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        sample = np.unique(np.concatenate([rng.integers(0, n_lines, size=24), [0, n_lines - 1, n_lines // 2]]))
        s_ids = np.concatenate([ids[offsets[i]:offsets[i + 1]] for i in sample]).astype(np.uint32)
        s_off = np.zeros(sample.size + 1, dtype=np.uint64); np.cumsum([offsets[i + 1] - offsets[i] for i in sample], out=s_off[1:])
        want = orc.embed_lines(table_h, s_ids, s_off, True, 2048)
        got = o[torch.from_numpy(sample).to(dev)].cpu().numpy()
        key = f"{kind} {dist}"
        out[key] = {"lines": n_lines, "tokens_pooled": used, "ms": round(ms, 3), "lines_per_s": round(n_lines / ms * 1e3), "tokens_per_s": round(used / ms * 1e3),
                    "algorithmic_GBps": round((used + n_lines) * 1024 / ms / 1e6, 1), "sample_bit_exact": bool(np.array_equal(got, want))}
        print(key, out[key], file=sys.stderr)
        del d_ids, d_off, o
print(json.dumps(out, indent=1))
