"""Where does a small host-form call's time go?  Wall us per call (median of 7 x 300), delivered answers against copy + synchronise,
and the kernels' own durations by HIP events.  Run on the GPU box: python tools/small_call_breakdown.py > gpurun_out/r06_small_calls.jsonl"""
import os, sys, time, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt
from semtools_amd import _lib as L
from tests import synth
ctx = smt.Context(0)
emb = synth.unit_rows(65536, seed=5)
qh = synth.unit_query(3, nq=4)
o_rows = np.empty((4, 16), dtype=np.uint64); o_dist = np.empty((4, 16), dtype=np.float64); o_cnt = np.zeros(4, dtype=np.uint64)
def call(c, nq, k):
    L.check(L.lib().smt_search(c._h, L.np_ptr(qh), nq, k, float("nan"), L.MODE_DOCUMENTS, None, 0, 0, L.np_ptr(o_rows), L.np_ptr(o_dist), L.np_ptr(o_cnt), 16))
def timed(c, nq, k, reps=300):
    for _ in range(30): call(c, nq, k)
    v = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(reps): call(c, nq, k)
        v.append((time.perf_counter() - t0) / reps * 1e6)
    return round(float(np.median(v)), 2)
out = []
for n in (1, 1000, 65536):
    c = smt.Corpus(ctx); c.append(emb[:n])
    for nq in (1, 2, 3, 4):
        row = {"rows": n, "nq": nq}
        for direct in (1, 0):
            ctx.set_tuning("direct_delivery", direct)
            ctx.prof_enable(False)
            row["delivered_us" if direct else "copy_sync_us"] = timed(c, nq, 3)
        ctx.prof_enable(True); ctx.prof_reset()
        for _ in range(50): call(c, nq, 3)
        for fam in ("scan", "select", "gemm"):
            nl, ms = ctx.prof_read(fam)
            if nl: row[fam + "_kernel_us_by_events"] = round(ms / nl * 1e3, 2)
        ctx.prof_enable(False)
        out.append(row)
        print(json.dumps(row), flush=True)
    c.close()
