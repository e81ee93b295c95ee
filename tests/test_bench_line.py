"""bench.py's LAST stdout line must survive the driver's 8 KB tail with every leg's figures in it: the full
per-leg objects go to a detail file, the line carries flat keys.  Checked here on a canned detail object (a real run's, with the
longest strings a leg can produce), no GPU."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _canned():
    with open(os.path.join(ROOT, "tests", "golden", "bench_detail_canned.json")) as f:
        return json.load(f)


def test_compact_line_fits_the_drivers_tail_and_carries_every_leg():
    import bench

    d = _canned()
    line = bench.compact_line(d)
    text = json.dumps(line)
    assert len(text) < 6000, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key
    assert len(line["cpu_baseline"]["sample"]) <= 120      # the driver's `parsed` cuts strings there
    for key in ("c3_queries_per_s", "c3_ms_per_batch", "c3_frac_of_2p5PF", "c3_image_queries_per_s", "c3_f32mfma_frac_of_157TF",
                "c4_rows_per_s", "c4_frac_hbm", "embed_frac_hbm_uniform", "ivf_recall_at_k", "ivf_queries_per_s",
                "ws_1q_frac_hbm", "ws_batch_queries_per_s", "ws_batch_cost_per_scanned_row_vs_unfiltered", "ingest_lines_per_s", "ingest_cores", "group_issue_us_8_logical_shards",
                "checks_ok", "checks_failed", "checks_total"):
        assert key in line, key
    assert line["checks_ok"] is True and line["checks_failed"] == []


def test_per_rank_figures_of_a_sharded_run_reach_the_line():
    """N > 1: the line must explain its own number -- every rank's scan / select microseconds, the merging device's wait for the other
    ranks' lists and its merge, the transport that ran, every rank's own rate -- and still fit the tail at 8 ranks."""
    import bench

    d = _canned()
    n = 8
    d["config"]["group"] = {"n_ranks": n, "n_local": 1, "first_rank": 0, "rccl_ranks": n, "rccl_version": 22606, "transport": "rccl",
                            "mode": "one rank per process (smt_group_create_rank: ncclCommInitRank)"}
    d["n_gpus"] = n
    d["ranks"] = {"process_model": "one rank per process", "transport": "rccl", "rccl_ranks": n, "n_ranks": n,
                  "scan_avg_us": [150.123456 + i for i in range(n)], "select_avg_us": [14.87654 + 0.1 * i for i in range(n)],
                  "exchange_wait_us": 31.4159, "merge_us": 4.2424, "exchange_wait_us_per_rank": [30.0 + i for i in range(n)],
                  "rank_rows_per_s": [6.2e9 - 1e7 * i for i in range(n)], "rank_elapsed_ms": [3.2 + 0.01 * i for i in range(n)], "note": "x" * 400}
    d["roofline"]["frac_per_rank"] = [0.84 - 0.005 * i for i in range(n)]
    line = bench.compact_line(d)
    rk = line["ranks"]
    assert rk["transport"] == "rccl" and rk["rccl_ranks"] == n and rk["process_model"] == "one rank per process"
    assert len(rk["scan_avg_us"]) == n and len(rk["select_avg_us"]) == n and len(rk["rank_rows_per_s"]) == n
    assert rk["scan_avg_us"][0] == 150.1 and rk["exchange_wait_us"] == 31.42 and rk["merge_us"] == 4.242   # 4 significant digits
    assert line["roofline"]["frac_is"] == "min over ranks"
    assert "note" not in rk and len(json.dumps(line)) < 6000, len(json.dumps(line))


def test_failed_checks_and_leg_errors_surface_in_the_line():
    import bench

    d = _canned()
    d["secondary"]["checks"]["k2_path_agreement"] = "999/1000"
    d["c4"]["checks"]["rows_match_fp64_topk"] = False
    d["embed"] = {"error": "RuntimeError('x' * 500)" + "y" * 500}
    d["checks"]["oracle_dist_max_abs_diff"] = 3e-5
    line = bench.compact_line(d)
    assert line["checks_ok"] is False
    joined = " ".join(line["checks_failed"])
    for needle in ("k2_path_agreement=999/1000", "rows_match_fp64_topk=False", "embed:", "oracle_dist_max_abs_diff"):
        assert needle in joined, (needle, joined)
    assert len(json.dumps(line)) < 6000


def test_launcher_argv_for_one_rank_per_process():
    """--ranks-per-process 1 with WORLD_SIZE unset: bench.py becomes this torch.distributed.run command (the form the driver itself
    uses for N > 1: one rank per GPU, rendezvous on 127.0.0.1) with its own arguments passed through."""
    import sys

    import bench

    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5", "--ranks-per-process", "1"]
    cmd = bench.launcher_argv(bench.parse_args(argv), argv, port=29555)
    assert cmd == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                   "--master-port", "29555", os.path.join(ROOT, "bench.py")] + argv


def test_a_run_that_cannot_start_prints_a_line_and_exits_zero():
    """`python3 bench.py --gpus 8 --steps 20 --warmup 5` as the driver types it, on a machine without 8 GPUs (this container has none):
    the last stdout line parses, value is null, the error and the visible device count are in it, exit code 0."""
    import subprocess
    import sys

    for extra, env in (([], {}), (["--ranks-per-process", "1"], {}), ([], {"WORLD_SIZE": "4", "RANK": "0"})):
        e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        e.update(env)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"] + extra,
                           capture_output=True, text=True, env=e, cwd=ROOT, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        res = json.loads(r.stdout.strip().split("\n")[-1])
        assert res["value"] is None and res["n_gpus"] == 8 and res["steps"] == 20 and res["warmup"] == 5
        assert isinstance(res["n_gpus_visible"], int) and res["error"]
        for key in ("metric", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
            assert key in res, key


def test_a_failure_after_start_up_prints_a_line_and_exits_one(monkeypatch, capsys):
    """Anything that goes wrong once the run has started (a library error in the timed loop of an N-GPU run nobody could try here) leaves
    the traceback on stderr AND a parseable last line with value null and the error -- exit code 1, because it is not an environment
    problem."""
    import bench

    def boom(args, under_launcher):
        raise RuntimeError("hipErrorLaunchFailure in step 3")

    monkeypatch.setattr(bench, "run", boom)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 8)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert bench.main(["--gpus", "8", "--steps", "20", "--warmup", "5"]) == 1
    out, err = capsys.readouterr()
    res = json.loads(out.strip().split("\n")[-1])
    assert res["value"] is None and res["n_gpus"] == 8 and "hipErrorLaunchFailure" in res["error"] and "failed after start-up" in res["error"]
    assert "Traceback" in err
