"""The less-travelled search forms across their parameters, ms per host call, each answer checked against fp64 torch:
 (B) threshold mode (every row under max_distance, src/search/mod.rs:115-116) across hit fractions at 1 M and 10 M rows;
 (C) top_k beyond the 56 of the list kernels (all-keys path) up to 10 000;
 (E) a corpus grown by many small appends.
python tools/sweep_modes.py > gpurun_out/sweep_modes.json"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import semtools_amd as smt

dev = torch.device("cuda", 0)
ctx = smt.Context(0)
g = torch.Generator(device=dev); g.manual_seed(3)
rows = 10_000_000
x = torch.empty((rows, 256), device=dev)
for b in range(0, rows, 2_000_000):
    c = torch.randn(2_000_000, 256, device=dev, generator=g); c /= c.norm(dim=1, keepdim=True); x[b:b + 2_000_000] = c
del c
g.manual_seed(4)
q = torch.randn(4, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
qh = np.ascontiguousarray(q.cpu().numpy())
out = {}

def t_call(fn, reps=3):
    fn(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    return (time.perf_counter() - t0) / reps * 1e3, r

for n in (1_000_000, 10_000_000):
    corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=n)
    d64 = 1.0 - (x[:n].double() @ q[0].double())
    srt = torch.sort(d64).values
    for frac in (0.0, 1e-6, 1e-4, 1e-2, 0.1, 0.5, 1.0):
        thr = float(srt[min(n - 1, int(frac * n))].item()) if frac < 1.0 else 2.5
        if frac == 0.0: thr = float(srt[0].item()) - 1e-3
        ms, got = t_call(lambda: corpus.search(qh[:1], top_k=3, max_distance=thr))
        rws, dst = got[0]
        want = int((d64 < thr).sum().item())
        ok = len(rws) == want and bool(np.all(np.diff(dst) >= 0)) and (want == 0 or abs(dst[0] - float(srt[0].item())) < 1e-12)
        out[f"threshold rows={n} hit_frac={frac}"] = {"ms": round(ms, 3), "hits": int(len(rws)), "hits_fp64": want, "ok": bool(ok)}
        print(f"threshold rows={n} hit_frac={frac}", out[f"threshold rows={n} hit_frac={frac}"], file=sys.stderr)
    for k in (56, 57, 64, 100, 1000, 10000):
        ms, got = t_call(lambda: corpus.search(qh[:1], top_k=k))
        v, i = torch.topk(d64, k, largest=False)
        ok = got[0][0].tolist() == i.cpu().numpy().tolist() or set(got[0][0].tolist()) == set(i.cpu().numpy().tolist())
        out[f"top_k rows={n} k={k}"] = {"ms": round(ms, 3), "ok": bool(ok), "max_abs_dist_diff": float(np.abs(got[0][1] - v.cpu().numpy()).max())}
        print(f"top_k rows={n} k={k}", out[f"top_k rows={n} k={k}"], file=sys.stderr)
    for nq, k in ((4, 100), (4, 1000)):
        ms, got = t_call(lambda: corpus.search(qh[:nq], top_k=k))
        out[f"top_k rows={n} k={k} nq={nq}"] = {"ms": round(ms, 3)}
        print(f"top_k rows={n} k={k} nq={nq}", out[f"top_k rows={n} k={k} nq={nq}"], file=sys.stderr)
    del d64, srt
    corpus.close()

# (E) many small appends
xh = x[:200_000].cpu().numpy()
for piece in (1, 10, 100, 10_000):
    c = smt.Corpus(ctx)
    n_app = min(20_000, 200_000 // piece)
    t0 = time.perf_counter()
    for a in range(n_app):
        c.append(xh[a * piece:(a + 1) * piece])
    ctx.synchronize()
    dt = time.perf_counter() - t0
    got = c.search(qh[:1], top_k=5)
    d = 1.0 - xh[:n_app * piece].astype(np.float64) @ qh[0].astype(np.float64)
    ok = got[0][0].tolist() == np.argsort(d, kind="stable")[:5].tolist()
    out[f"append piece={piece}"] = {"appends": n_app, "us_per_append": round(dt / n_app * 1e6, 2), "rows_per_s": round(n_app * piece / dt), "search_ok": bool(ok)}
    print(f"append piece={piece}", out[f"append piece={piece}"], file=sys.stderr)
    c.close()
print(json.dumps(out, indent=1))
