"""c3 (1000 queries x 10 M rows) across top_k and query counts: ms per host-form call, the number of queries the exactness certificate
sent to the exhaustive re-answer (uncertain), with and without the operand image.  A cliff here is a candidate-buffer overflow or a
guard band too narrow for the k -- answers stay exact either way.  python tools/sweep_c3_k.py > gpurun_out/sweep_c3_k.json"""
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
g.manual_seed(5)
q = torch.randn(1000, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True); q = np.ascontiguousarray(q.cpu().numpy())
corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
out = {}
for image in (False, True):
    if image: corpus.prepack()
    for nq in (1000, 100, 16):
        for k in (1, 3, 10, 30, 48, 56):
            ctx.uncertain_count()
            corpus.search(q[:nq], top_k=k)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): got = corpus.search(q[:nq], top_k=k)
            ms = (time.perf_counter() - t0) / 3 * 1e3
            unc = int(ctx.uncertain_count())
            # fp64 check of one query
            d = 1.0 - (x.double() @ torch.from_numpy(q[nq - 1]).to(dev).double())
            v, i = torch.topk(d, k, largest=False)
            ok = set(got[nq - 1][0].tolist()) == set(i.cpu().numpy().tolist())
            del d
            out[f"image={int(image)} nq={nq} k={k}"] = {"ms": round(ms, 3), "uncertain_in_4_calls": unc, "last_query_matches_fp64": bool(ok)}
            print(f"image={int(image)} nq={nq} k={k}", out[f"image={int(image)} nq={nq} k={k}"], file=sys.stderr)
print(json.dumps(out, indent=1))
