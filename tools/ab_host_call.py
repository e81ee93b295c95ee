"""A/B of the host-form call's fixed costs (tuning keys host_direct, gemm_flat_small): wall time per smt_search call, answers compared
across the settings.  Both keys were measured and NOT kept: the library ignores unknown keys with an error, so this
script runs only against the experiment's build.  Run on the GPU box: python tools/ab_host_call.py > gpurun_out/ab_host_call.json"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import semtools_amd as smt
from semtools_amd import _lib as L

dev = torch.device("cuda", 0)
ctx = smt.Context(0)
g = torch.Generator(device=dev); g.manual_seed(3)
rows = 10_000_000
x = torch.empty((rows, 256), device=dev)
for b in range(0, rows, 2_000_000):
    c = torch.randn(2_000_000, 256, device=dev, generator=g); c /= c.norm(dim=1, keepdim=True); x[b:b + 2_000_000] = c
del c
q = torch.randn(32, 256, device=dev, generator=g); q /= q.norm(dim=1, keepdim=True); q = np.ascontiguousarray(q.cpu().numpy())
per = rows // 10_000
packed = smt.PackedRanges([(d * per, (d + 1) * per) for d in range(0, 10_000, 2)])
k = 10
o_rows = np.empty((32, k), dtype=np.uint64); o_dist = np.empty((32, k), dtype=np.float64); o_cnt = np.zeros(32, dtype=np.uint64)

def call(corpus, nq, filtered, mode):
    L.check(L.lib().smt_search(corpus._h, L.np_ptr(q), nq, k, 0.9 if mode == smt.MODE_WORKSPACE else float("nan"), mode,
                               C.cast(packed.arr, C.c_void_p) if filtered else None, packed.n if filtered else 0, 0,
                               L.np_ptr(o_rows), L.np_ptr(o_dist), L.np_ptr(o_cnt), k))
    return o_rows[:nq].copy(), o_dist[:nq].copy(), o_cnt[:nq].copy()

def timed(corpus, nq, filtered, mode, reps):
    for _ in range(3): got = call(corpus, nq, filtered, mode)
    best = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps): call(corpus, nq, filtered, mode)
        best.append((time.perf_counter() - t0) / reps * 1e6)
    return float(np.median(best)), got

big = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
mid = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=1_000_000)
tiny = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=4096)
big.prepack()
cases = [("tiny_4096_rows_1q", tiny, 1, False, smt.MODE_DOCUMENTS, 200), ("c2_1M_rows_1q_f32", mid, 1, False, smt.MODE_DOCUMENTS, 100),
         ("ws_1q_image_5M_of_10M", big, 1, True, smt.MODE_WORKSPACE, 40), ("1q_image_10M", big, 1, False, smt.MODE_WORKSPACE, 40),
         ("8q_image_10M", big, 8, False, smt.MODE_WORKSPACE, 20), ("ws_8q_image_5M_of_10M", big, 8, True, smt.MODE_WORKSPACE, 20),
         ("32q_image_10M", big, 32, False, smt.MODE_WORKSPACE, 20)]
out = {}
ref = {}
for hd, fs in ((0, 0), (1, 0), (0, 1), (1, 1)):
    ctx.set_tuning("host_direct", hd); ctx.set_tuning("gemm_flat_small", fs)
    ctx.uncertain_count()
    row = {}
    for name, corpus, nq, filt, mode, reps in cases:
        us, got = timed(corpus, nq, filt, mode, reps)
        row[name] = round(us, 1)
        if name not in ref: ref[name] = got
        else:
            same = all(np.array_equal(a, b) for a, b in zip(ref[name], got))
            if not same: row[name + "_DIFFERS"] = True
    row["uncertain"] = int(ctx.uncertain_count())
    out[f"host_direct={hd} gemm_flat_small={fs}"] = row
    print(f"host_direct={hd} flat={fs}", row, file=sys.stderr)
print(json.dumps(out, indent=1))
