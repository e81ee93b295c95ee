"""smt_sharded_search (host form) on 1 / 2 / 4 / 8 logical shards of ONE GPU against the unsharded corpus of the same 8 M rows: ms per
call across query counts and search forms.  The device work serialises on one GPU here, so the ratio to the unsharded call is an
upper bound of what the group layer adds (exchange, merge, per-shard fixed costs); answers are compared with the unsharded ones.
python tools/sweep_sharded.py > gpurun_out/sweep_sharded.json"""
import gc, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import semtools_amd as smt

gc.disable()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
rows = 8_000_000
x = torch.randn(rows, 256, device=dev, generator=g); x /= x.norm(dim=1, keepdim=True)
g.manual_seed(5)
q = torch.randn(1000, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True)
qh = np.ascontiguousarray(q.cpu().numpy())
docs = [(d * 1000, (d + 1) * 1000) for d in range(0, rows // 1000, 2)]
cases = [("top10", dict(top_k=10)), ("top56", dict(top_k=56)), ("top100", dict(top_k=100)), ("thr_0.85", dict(top_k=3, max_distance=0.85)),
         ("ws_subset", dict(top_k=10, max_distance=0.9, mode=smt.MODE_WORKSPACE, ranges=docs))]
out = {}
ref = {}
for n_sh in (0, 1, 2, 4, 8):
    if n_sh == 0:
        ctx = smt.Context(0)
        corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
        sync = ctx.synchronize
    else:
        grp = smt.Group.logical(0, n_sh)
        per = rows // n_sh
        corpus = smt.ShardedCorpus(grp, device_ptrs=[x.data_ptr() + i * per * 1024 for i in range(n_sh)], shard_rows=[per] * n_sh)
        sync = grp.synchronize
    for name, kw in cases:
        for nq in (1, 16, 256, 1000):
            if name in ("top100", "thr_0.85") and nq > 16: continue
            kw2 = dict(kw)
            if "ranges" in kw2: kw2["ranges"] = smt.PackedRanges(docs) if n_sh == 0 else docs
            got = corpus.search(qh[:nq], **kw2); corpus.search(qh[:nq], **kw2); sync()
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps): got = corpus.search(qh[:nq], **kw2)
            ms = (time.perf_counter() - t0) / reps * 1e3
            key = f"{name} nq={nq}"
            if n_sh == 0: ref[key] = got
            same = all(a[0].tolist() == b[0].tolist() and np.array_equal(a[1], b[1]) for a, b in zip(got, ref[key]))
            out.setdefault(key, {})["unsharded" if n_sh == 0 else f"{n_sh}_shards"] = round(ms, 3)
            if not same: out[key][f"{n_sh}_shards_DIFFERS"] = True
    corpus.close()
    if n_sh: grp.close()
    print("done", n_sh, file=sys.stderr)
for k, v in out.items(): print(k, v, file=sys.stderr)
print(json.dumps(out, indent=1))
