"""Wave timeline of one block of gemm_rowreg_kernel<2> (trace build: -DSMT_RR_EXP=256, tools/exp_k3_trace.sh): s_memtime stamps
of block 40's second step at 1000 queries x 10 M rows.  Prints, per wave, the phases of the row phase and the mean / max cycles of
each phase of a sweep position."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt  # noqa: E402

NAMES = {1: "step_start", 2: "loads_issued", 3: "rows_arrived_norms", 4: "converted", 5: "product_start", 6: "mfmas_issued",
         7: "epilogue_done", 11: "slow_enter", 12: "slow_slots", 13: "slow_written", 8: "own_dma_landed", 9: "barrier_passed"}


def main():
    rows, nq = 10_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.empty(rows, 256, device=dev)
    for b in range(0, rows, 2_000_000):
        c = torch.randn(min(2_000_000, rows - b), 256, device=dev, generator=g)
        x[b:b + c.shape[0]] = c / c.norm(dim=1, keepdim=True)
    q = torch.randn(nq, 256, device=dev, generator=g)
    stamps = torch.zeros(4 << 20, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
    if os.environ.get("TRACE_IMAGE", "1") != "0":
        corpus.prepack()
    out_rows = torch.empty(nq, 10, dtype=torch.int64, device=dev)
    out_dist = torch.empty(nq, 10, dtype=torch.float64, device=dev)
    for _ in range(2):
        corpus.search_topk_device(q.data_ptr(), nq, 10, 0, out_rows.data_ptr(), out_dist.data_ptr())
    ctx.synchronize()
    ctx.set_tuning("scan_debug_ptr", stamps.data_ptr())
    corpus.search_topk_device(q.data_ptr(), nq, 10, 0, out_rows.data_ptr(), out_dist.data_ptr())
    ctx.synchronize()
    s = stamps[:8 * 1024].cpu().numpy().astype(np.uint64).reshape(8, 1024)
    ck = stamps[8 * 1024:8 * 1024 + 4].cpu().numpy().astype(np.int64)
    out = {"nq": nq, "waves": {}, "shader_clock_GHz_over_8_steps": round(float(ck[2] - ck[0]) / float(ck[3] - ck[1]) * 0.1, 3) if ck[3] > ck[1] else None}
    for w in range(8):
        v = s[w][s[w] != 0]
        ids = (v & np.uint64(255)).astype(int)
        t = (v >> np.uint64(8)).astype(np.int64)
        if len(t) == 0:
            continue
        t0 = t[0]
        ev = [(int(i), int(tt - t0)) for i, tt in zip(ids, t)]
        out.setdefault("barrier_arrive", {})[w] = [int(tt + t0 - int((s[0][0] >> np.uint64(8)))) for i, tt in ev if i == 8]
        out.setdefault("barrier_release", {})[w] = [int(tt + t0 - int((s[0][0] >> np.uint64(8)))) for i, tt in ev if i == 9]
        out.setdefault("product_start", {})[w] = [int(tt + t0 - int((s[0][0] >> np.uint64(8)))) for i, tt in ev if i == 5]
        trans = {}
        for (i0, a0), (i1, a1) in zip(ev[:-1], ev[1:]):
            trans.setdefault(f"{NAMES.get(i0, i0)}->{NAMES.get(i1, i1)}", []).append(a1 - a0)
        out["waves"][w] = {"total_ticks": ev[-1][1], "transitions": {k: {"n": len(a), "mean": round(float(np.mean(a)), 1), "max": int(max(a)), "sum": int(sum(a))} for k, a in trans.items()},
                           "first_events": ev[:24]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
