"""GPU: the SHIPPED defaults at BASELINE.json's sizes, in the driver-run suite (VERDICT r2 "next" 2):
  * c3 -- 1000 batched queries x 10 M rows in the default nomination mode (f16 x 2 from 128 queries on): every query against
    the single-query scan path, a sample against an independent fp64 evaluation, zero selects without certificate;
  * c4-size shards -- returned ROW INDICES (not only distances) at 32 M rows on one GPU and on a 3-shard group against a
    chunked fp64 top-k;
  * bench.py's N > 1 code path (library-side RCCL all-gather + device merge) under torchrun on one rank;
  * the MFMA rounding probe whose outcome the certificate's error bounds rest on."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unit(rows, seed, chunk=2_000_000):
    import torch

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.empty((rows, 256), device=dev, dtype=torch.float32)
    for b in range(0, rows, chunk):
        e = min(rows, b + chunk)
        c = torch.randn(e - b, 256, device=dev, generator=g)
        c /= c.norm(dim=1, keepdim=True)
        x[b:e] = c
    return x


def _fp64_topk(x, q, k, chunk=2_000_000):
    """(distances, rows) of the k smallest fp64 cosine distances, chunked (x.double() of 32 M rows does not fit)."""
    import torch

    best_v = best_i = None
    qd = q.double()
    for b in range(0, x.shape[0], chunk):
        d = 1.0 - (x[b:b + chunk].double() @ qd) / (x[b:b + chunk].double().norm(dim=1) * qd.norm())
        v, i = torch.topk(d, min(k, d.numel()), largest=False)
        i = i + b
        if best_v is not None:
            v, i = torch.cat([best_v, v]), torch.cat([best_i, i])
            v, sel = torch.topk(v, min(k, v.numel()), largest=False)
            i = i[sel]
        best_v, best_i = v, i
    return best_v.cpu().numpy(), best_i.cpu().numpy()


def test_c3_thousand_queries_ten_million_rows_in_the_default_mode(gpu_ctx):
    """What bench.py's c3 leg times, as a test: the default dispatch (no tuning key touched) on 1000 x 10 M."""
    import torch
    import semtools_amd as smt

    rows, nq, k = 10_000_000, 1000, 10
    x = _unit(rows, 3)
    q = _unit(nq, 5)
    x[1_234_567] = q[0]
    x[[9_999_999, 5]] = q[1]
    torch.cuda.synchronize()
    for key, default in (("gemm_bf16x3", 1), ("gemm_rowreg", 1), ("gemm_nominate", 0), ("guard_band", 8)):
        gpu_ctx.set_tuning(key, default)                     # (other test modules restore these; make sure)
    c = smt.Corpus(gpu_ctx, device_ptr=x.data_ptr(), rows=rows)
    dev = x.device
    out_rows = torch.empty(nq, k, dtype=torch.int64, device=dev)
    out_dist = torch.empty(nq, k, dtype=torch.float64, device=dev)
    gpu_ctx.uncertain_count()
    gpu_ctx.prof_enable(True)
    gpu_ctx.prof_reset()
    c.search_topk_device(q.data_ptr(), nq, k, 0, out_rows.data_ptr(), out_dist.data_ptr())      # K3, nominated with f16 x 1 (the default from 256 queries)
    torch.cuda.synchronize()
    launches, _ = gpu_ctx.prof_read("gemm")
    gpu_ctx.prof_enable(False)
    assert launches > 0                                       # the batched kernels ran
    assert gpu_ctx.uncertain_count() == 0                     # every select carried its exactness certificate
    assert out_rows[0, 0].item() == 1_234_567 and out_dist[0, 0].item() < 2.3e-16
    assert out_rows[1, :2].tolist() == [5, 9_999_999]
    # every query against the single-query scan path (K2, <= 4 queries per pass): rows and f64 distances identical
    k2_rows, k2_dist = torch.empty_like(out_rows), torch.empty_like(out_dist)
    gpu_ctx.set_tuning("gemm_min_nq", 8)                     # (4 queries over 10 M rows would take K3 too: keep them on the scan kernel)
    gpu_ctx.prof_enable(True)
    gpu_ctx.prof_reset()
    try:
        for i in range(0, nq, 4):
            c.search_topk_device(q[i:i + 4].data_ptr(), 4, k, 0, k2_rows[i:i + 4].data_ptr(), k2_dist[i:i + 4].data_ptr())
        torch.cuda.synchronize()
    finally:
        gpu_ctx.set_tuning("gemm_min_nq", 5)
    assert gpu_ctx.prof_read("gemm")[0] == 0                  # no batched kernel in the truth
    gpu_ctx.prof_enable(False)
    assert bool((k2_rows == out_rows).all()) and bool((k2_dist == out_dist).all())
    # ... and the same batch over the corpus' fp16 operand image (smt_corpus_prepack): identical again
    c.prepack()
    img_rows, img_dist = torch.empty_like(out_rows), torch.empty_like(out_dist)
    c.search_topk_device(q.data_ptr(), nq, k, 0, img_rows.data_ptr(), img_dist.data_ptr())
    torch.cuda.synchronize()
    assert gpu_ctx.uncertain_count() == 0
    assert bool((img_rows == out_rows).all()) and bool((img_dist == out_dist).all())
    # 16 queries against an independent fp64 evaluation: indices and distances
    for i in list(range(8)) + [100, 257, 400, 511, 640, 777, 901, 999]:
        tv, ti = _fp64_topk(x, q[i], k)
        assert out_rows[i].cpu().tolist() == ti.tolist(), i
        np.testing.assert_allclose(out_dist[i].cpu().numpy(), np.maximum(tv, 0.0), rtol=0, atol=1e-9)
    # the host form (smt_search) of the same batch agrees with the device form
    got = c.search(q[:160].cpu().numpy(), top_k=k)
    for i in range(160):
        assert got[i][0].tolist() == out_rows[i].cpu().tolist()
    c.close()


def test_c4_size_shards_return_the_right_rows(gpu_ctx):
    """32 M rows (32.8 GB): one GPU, and the same rows as three adopted shards of a logical group -- host form and
    device-resident form (what bench.py --gpus N times) -- against a chunked fp64 top-k: ROW INDICES and distances."""
    import torch
    import semtools_amd as smt

    rows, k = 32_000_000, 10
    x = _unit(rows, 3)
    qs = _unit(3, 4)
    x[[31_999_999, 12, 20_000_000]] = qs[0]                  # exact ties across what will be shard borders
    torch.cuda.synchronize()
    truth = [_fp64_topk(x, qs[i], k) for i in range(3)]
    assert sorted(truth[0][1][:3].tolist()) == [12, 20_000_000, 31_999_999]
    want0 = [12, 20_000_000, 31_999_999] + truth[0][1][3:].tolist()      # ties in row order, then the rest
    qh = qs.cpu().numpy()

    def check(got):
        assert got[0][0].tolist() == want0
        for i in (1, 2):
            assert got[i][0].tolist() == truth[i][1].tolist(), i
        for i in range(3):
            np.testing.assert_allclose(got[i][1], np.maximum(np.sort(truth[i][0]), 0.0), rtol=0, atol=1e-9)

    one = smt.Corpus(gpu_ctx, device_ptr=x.data_ptr(), rows=rows)
    check([one.search(qh[i], top_k=k)[0] for i in range(3)])            # single queries: K2
    check(one.search(qh, top_k=k))                                       # 3 queries on a large shard: K3
    one.close()
    g = smt.Group.logical(0, 3)
    cuts = [0, 11_000_000, 20_000_001, rows]                             # unequal shards, a tie row on each side of a border
    ptrs = [x[cuts[i]:cuts[i + 1]].data_ptr() for i in range(3)]
    sc = smt.ShardedCorpus(g, device_ptrs=ptrs, shard_rows=[cuts[i + 1] - cuts[i] for i in range(3)])
    check([sc.search(qh[i], top_k=k)[0] for i in range(3)])
    check(sc.search(qh, top_k=k))
    # device-resident form: one packed [2][k] answer per query on local device 0
    out = torch.empty((3, 2, k), dtype=torch.int64, device=x.device)
    for i in range(3):
        sc.search_topk_device([qs[i].data_ptr()] * 3, 1, k, [out[i].data_ptr(), None, None])
    g.synchronize()
    o = out.cpu().numpy()
    check([(o[i, 0].view(np.uint64), np.ascontiguousarray(o[i, 1]).view(np.float64)) for i in range(3)])
    sc.close()
    g.close()


def test_bench_exchange_path_under_torchrun_on_one_rank():
    """bench.py's N > 1 code path -- the ranks join one smt_group (ncclCommInitRank), scan + select + ncclAllGather +
    merge inside the library, c4 sharded -- launched the way the driver launches it, with one rank (all a 1-GPU box has):
    the JSON line is the last thing on stdout, carries the contract's keys and passes its own checks."""
    env = dict(os.environ, SEMTOOLS_BENCH_FORCE_EXCHANGE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--settle-steps", "8",
           "--no-secondary", "--no-ivfpq", "--no-embed", "--no-cpu-baseline", "--no-workspace", "--no-ingest", "--no-group-issue", "--c4-rows", "4000000",
           "--c4-steps", "4", "--detail-out", os.path.join(ROOT, "gpurun_out", "bench_detail_forced_exchange.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().split("\n")[-1]
    res = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in res, key
    assert res["n_gpus"] == 1 and res["steps"] == 5 and res["value"] > 0
    assert len(line) < 6000, len(line)                     # the driver keeps an 8 KB tail of stdout
    assert res["config"]["forced_exchange_on_one_rank"] is True and res["config"]["group"]["rccl_ranks"] == 1
    assert res["config"]["group"]["transport"] == "rccl" and "ncclCommInitRank" in res["config"]["group"]["mode"]
    assert res["checks_ok"] is True and res["checks_failed"] == [] and res["checks_total"] >= 5, res
    assert res["c4_rows_per_s"] > 0 and 0 < res["c4_frac_hbm"] < 1.0
    assert res["roofline"]["bound"] == "hbm" and 0 < res["roofline"]["frac"] < 1.0
    detail = json.load(open(os.path.join(ROOT, res["detail_file"])))   # the full per-leg objects
    assert detail["config"]["group"]["rccl_ranks"] == 1
    assert detail["checks"]["torch_fp64_topk_distances_match"] is True
    assert detail["c4"]["checks"]["torch_fp64_topk_distances_match"] is True and detail["c4"]["checks"]["rows_match_fp64_topk"] is True
    assert "rccl_ranks=1" in r.stderr                      # the pre-run diagnostics of a multi-rank launch
    rk = res["ranks"]                                      # (the per-rank figures travel through dist.all_gather_object in this process model)
    assert rk["process_model"] == "one rank per process" and rk["transport"] == "rccl" and rk["rccl_ranks"] == 1
    assert len(rk["scan_avg_us"]) == 1 and rk["scan_avg_us"][0] > 0 and rk["select_avg_us"][0] > 0 and rk["exchange_wait_us"] > 0 and rk["merge_us"] > 0
    assert detail["checks"]["every_timed_answer_proved_exact"] == "5/5"


SMALL = ["--steps", "5", "--warmup", "2", "--settle-steps", "8", "--no-secondary", "--no-ivfpq", "--no-embed", "--no-cpu-baseline",
         "--no-workspace", "--no-ingest", "--no-group-issue", "--c4-rows", "4000000", "--c4-steps", "4"]


def _bench(args, timeout=600, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, cwd=ROOT, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    line = r.stdout.strip().split("\n")[-1]
    return r, json.loads(line), line


@pytest.mark.parametrize("transport", [None, "rccl"])
def test_bench_one_process_group_path_on_one_gpu(transport):
    """`python bench.py --gpus N` with WORLD_SIZE unset = ONE process drives the N GPUs through smt_group_create (ncclCommInitAll), the
    reference's process model.  What a 1-GPU box runs of it: --gpus 1 --single-process -- the same code with one device -- with the default
    transport (peer reads) and with the RCCL all-gather."""
    args = ["--gpus", "1", "--single-process"] + SMALL + ["--detail-out", os.path.join(ROOT, "gpurun_out", "bench_detail_one_process.json")]
    if transport:
        args += ["--group-transport", transport]
    r, res, line = _bench(args)
    assert r.returncode == 0, r.stderr[-2000:]
    assert res["n_gpus"] == 1 and res["value"] > 0 and len(line) < 6000
    grp = res["config"]["group"]
    assert grp["n_ranks"] == 1 and grp["rccl_ranks"] == 1 and grp["transport"] == (transport or "peer") and "ncclCommInitAll" in grp["mode"]
    assert res["checks_ok"] is True and res["checks_failed"] == [] and res["checks_total"] >= 6, res
    assert res["c4_rows_per_s"] > 0 and 0 < res["c4_frac_hbm"] < 1.0 and 0 < res["roofline"]["frac"] < 1.0


def test_bench_one_process_over_logical_shards():
    """The N > 1 branch of the one-process bench (per-device shards and queries, one answer on device 0, c4 cut over the shards,
    the fp64 check over every shard) on 3 logical ranks of the one GPU."""
    args = [a for a in SMALL if a != "--no-ivfpq"] + ["--c5-rows-total", "300000"]      # ... and the sharded c5 leg at a small size
    r, res, line = _bench(["--logical-shards", "3"] + args + ["--detail-out", os.path.join(ROOT, "gpurun_out", "bench_detail_logical.json")])
    assert r.returncode == 0, r.stderr[-2000:]
    assert res["ivf_sharded_rows_total"] == 300000 and res["ivf_sharded_recall_at_k"] >= 0.9 and res["ivf_sharded_queries_per_s"] > 0, res
    assert res["n_gpus"] == 1 and res["config"]["logical_shards_on_one_gpu"] == 3 and res["config"]["group"]["n_ranks"] == 3
    assert res["config"]["group"]["transport"] == "peer" and res["config"]["group"]["rccl_ranks"] == 0
    assert res["checks_ok"] is True and res["checks_failed"] == [], res
    detail = json.load(open(os.path.join(ROOT, res["detail_file"])))
    assert detail["checks"]["rows_match_fp64_topk_over_all_shards"] is True
    assert detail["c4"]["checks"]["rows_match_fp64_topk"] is True and detail["c4"]["config"]["rows_per_gpu"] == 1333334
    assert abs(res["value"] - 3 * 1_000_000 * 5 / (res["ms_per_step"] * 5e-3)) / res["value"] < 1e-3   # rows of ALL shards / time
    # what makes a SCALE record diagnosable: every rank's scan / select time, the merging device's wait for the others + its merge, the
    # transport that really ran, every rank's own rate -- in the compact line; roofline.frac is the SLOWEST rank's
    rk = res["ranks"]
    assert rk["process_model"] == "one process, logical ranks" and rk["transport"] == "peer" and rk["rccl_ranks"] == 0
    for key in ("scan_avg_us", "select_avg_us", "rank_rows_per_s"):
        assert len(rk[key]) == 3 and all(v and v > 0 for v in rk[key]), (key, rk)
    assert rk["exchange_wait_us"] > 0 and 0 < rk["merge_us"] < 100
    assert res["roofline"]["frac_is"] == "min over ranks"
    assert abs(res["roofline"]["avg_kernel_us"] - max(rk["scan_avg_us"])) / res["roofline"]["avg_kernel_us"] < 0.02
    assert len(detail["roofline"]["frac_per_rank"]) == 3 and abs(min(detail["roofline"]["frac_per_rank"]) - detail["roofline"]["frac"]) < 1e-9
    assert detail["checks"]["every_timed_answer_proved_exact"] == "5/5"


def test_bench_more_gpus_than_the_box_has_is_a_line_not_a_traceback():
    """--gpus 2 on a 1-GPU box, typed as the driver types it (no torch.distributed.run): a parseable line with value null, the error and
    the number of visible devices; exit code 0."""
    for extra in ([], ["--ranks-per-process", "1"]):
        r, res, _ = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5"] + extra, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        assert res["value"] is None and res["n_gpus"] == 2 and res["n_gpus_visible"] == 1 and "visible" in res["error"]
        for key in ("metric", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
            assert key in res, key


def test_bench_abandons_a_c4_leg_that_hangs_and_still_prints_the_line():
    """With several ranks c4 is collective: a rank that fails inside it would leave the others in ncclAllGather for ever and the c2
    figures, complete by then, unprinted.  bench.py's watchdog abandons the leg after --c4-timeout seconds.  Driven here on one rank
    with a leg that sleeps for ever (SEMTOOLS_BENCH_FAKE_C4_HANG): exit code 0, the JSON line last on stdout with the c2 figures,
    the abandoned leg listed among the failed checks."""
    env = dict(os.environ, SEMTOOLS_BENCH_FAKE_C4_HANG="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--settle-steps", "8", "--no-secondary", "--no-ivfpq",
           "--no-embed", "--no-cpu-baseline", "--no-workspace", "--no-ingest", "--no-group-issue", "--c4-timeout", "3",
           "--detail-out", os.path.join(ROOT, "gpurun_out", "bench_detail_c4_abandoned.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().split("\n")[-1])
    assert res["value"] > 0 and 0 < res["roofline"]["frac"] < 1.0
    assert "c4_rows_per_s" not in res and res["checks_ok"] is False and any("c4" in c for c in res["checks_failed"]), res
    assert "leg abandoned" in r.stderr


def test_mfma_accumulate_rounding_probe():
    """tools/micro/mfma_rounding: the certificate's error bounds (common.h F32_ERR_*) assume that every MFMA instruction adds
    its K products to the accumulator with ONE rounding to nearest.  If this part ever rounds differently the bounds must be
    re-derived: fail loudly here."""
    exe = os.path.join(ROOT, "tools", "micro", "mfma_rounding")
    if not os.path.exists(exe):
        src = exe + ".hip"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    res = json.loads(r.stdout)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "mfma_rounding.json"), "w") as f:
        f.write(r.stdout)
    for name, k_adds in (("v_mfma_f32_32x32x2_f32", 128), ("v_mfma_f32_32x32x16_bf16", 16), ("v_mfma_f32_32x32x16_f16", 16)):
        m = res[name]
        print(name, m["accumulate_rounding"], m["dot256_positive"])
        # what DESIGN.md section 5 states and common.h's bounds use: nearest, at most one rounding per instruction
        assert m["accumulate_rounding"].startswith("round-to-nearest"), (name, m)
        assert m["dot256_positive"]["max_abs_error_in_units_of_2^-24_relative"] <= k_adds, (name, m)


def test_wave_sum4_reduction_tree_on_the_device():
    """tools/micro/wave_sum4: the scan kernel reduces the four rows of a chunk with one transposing tree (device_utils.h wave_sum4:
    quad permutes, row rotations, v_permlane16_swap / v_permlane32_swap through inline asm).  Lane l must hold the 64-lane sum of
    input l % 4 -- checked with integer-valued floats against host sums."""
    exe = os.path.join(ROOT, "tools", "micro", "wave_sum4")
    if not os.path.exists(exe):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                               exe + ".hip", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "PASS wave_sum4" in r.stdout, r.stdout + r.stderr
