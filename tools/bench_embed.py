"""K1 throughput on the GPU box: n_lines ragged lines over a V x 256 table, everything resident.
Reports lines/s, tokens/s and the gathered-bytes rate (algorithmic bytes = tokens*1 KiB + lines*1 KiB)."""
import argparse
import json
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt  # noqa: E402
from tests import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=2_000_000)
    ap.add_argument("--vocab", type=int, default=500_000)
    ap.add_argument("--max-tok", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--uniform", action="store_true", help="uniform token ids (no hot rows: every gather goes to HBM)")
    ap.add_argument("--tune", action="append", default=[], help="key=value for smt_set_tuning (repeatable)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    table = torch.randn(args.vocab, 256, device=dev) * 0.1
    ids, offsets = synth.token_lines(args.lines, V=args.vocab, seed=1, min_tok=0, max_tok=args.max_tok)
    if args.uniform:
        ids = np.random.default_rng(7).integers(0, args.vocab, size=ids.size).astype(np.uint32)
    d_ids = torch.from_numpy(ids.astype(np.int32)).to(dev)
    d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    out = torch.empty(args.lines, 256, device=dev)
    ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    model = smt.Model(ctx, device_ptr=table.data_ptr(), V=args.vocab, normalize=True)
    for kv in args.tune:
        key, val = kv.split("=")
        ctx.set_tuning(key, int(val))
    ctx.prof_enable(True)
    model.embed_device(d_ids.data_ptr(), d_off.data_ptr(), args.lines, 2048, out.data_ptr())
    ctx.synchronize()
    ctx.prof_reset()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        model.embed_device(d_ids.data_ptr(), d_off.data_ptr(), args.lines, 2048, out.data_ptr())
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / args.reps
    n, ms = ctx.prof_read("embed")
    ker = ms / max(n, 1) * 1e-3
    T = int(ids.size)
    alg_bytes = T * 1024 + args.lines * 1024
    # spot check against torch (fp32 sums in a different order -> tolerance, exactness is tested in tests/)
    i = 12345 % args.lines
    ref = table[torch.from_numpy(ids[int(offsets[i]):int(offsets[i + 1])].astype(np.int64)).to(dev)].sum(0)
    ref = ref / max(int(offsets[i + 1] - offsets[i]), 1)
    ref = ref / ref.norm().clamp_min(1e-12)
    print(json.dumps(dict(tune=args.tune, lines=args.lines, tokens=T, vocab=args.vocab, uniform_ids=bool(args.uniform), kernel_ms=round(ker * 1e3, 3),
                          wall_ms=round(wall * 1e3, 3), lines_per_s=round(args.lines / ker / 1e6, 1),
                          tokens_per_s_G=round(T / ker / 1e9, 2), gather_GBps=round(alg_bytes / ker / 1e9, 1),
                          frac_of_8TBps=round(alg_bytes / ker / 8e12, 3),
                          spot_max_abs_diff=float((out[i] - ref).abs().max()))))


if __name__ == "__main__":
    main()
