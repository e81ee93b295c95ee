#!/usr/bin/env python3
"""bench.py -- headline benchmark of the semtools search hot path on MI355X.

Metric (BASELINE.json): chunk-vectors scanned/sec (whole job).  Workload at N=1:
config c2 = 1 query x 1M chunks, f32, brute-force cosine, top-k on one MI355X
(HBM-bound single-vector path).  For N>1 the corpus is row-sharded, every rank
scans ITS 1M-row shard (weak scaling: `value`), and the per-shard top-k lists meet
in the path's one exchange step INSIDE the library
(smt_sharded_search_topk_device, csrc/group.cpp): in a one-process group the merge
kernel of device 0 reads the ranks' lists in place over xGMI (peer transport: one event
per rank, no gather); between processes (torchrun: ncclCommInitRank) it is one RCCL
all-gather of the packed lists + the merge.
Every N also runs BASELINE config c4 -- 1 query x 100M chunks row-sharded over the
N GPUs, 100M / N rows per GPU -- reported in the "c4" object of the same line.

A step = one query end to end: f32 scan of the resident shard -> hierarchical
top-k merge -> exact f64 rescoring -> (all-gather + merge when N>1) -> async
copy of the k (row, distance) pairs to pinned host memory.  Inputs (corpus,
queries) are resident in HBM before the timed region starts; steps are enqueued
back to back on one stream and the region ends with a full synchronise.

Launch: python bench.py [--gpus N --steps K --warmup W] starts by itself for every N:
  * WORLD_SIZE unset (the plain command line): ONE process drives the N GPUs -- the reference's process model
    (src/bin/semtools.rs:134-135: one synchronous task) and the one the Rust wrappers use (rust/src/search/hip.rs:
    smt_group_create -> ncclCommInitAll, peer access between the devices): one smt_group, one ShardedCorpus over N device
    shards, one issuing thread per device inside the library.  --ranks-per-process 1 re-executes the same command under
    torch.distributed.run (one rank per process, ncclCommInitRank);
  * WORLD_SIZE set (launched by torch.distributed.run / the driver): one rank per process, as before.
Fewer than N devices visible, or a group that cannot be created, is reported as a JSON line with "value": null, "error" and
"n_gpus_visible", exit code 0 -- never a traceback without a line.

Output: the LAST stdout line is one compact JSON object (< 6 KB: the driver keeps an 8 KB tail) -- the contract keys, the c2
`roofline` and `cpu_baseline`, and every other leg's figures hoisted to flat top-level keys (c3_*, c4_*, ws_*, embed_*, ingest_*,
ivf_*) plus `checks_ok` / `checks_failed`.  The full per-leg objects (notes, sources, per-stage tables) go to
gpurun_out/bench_detail.json (or --detail-out) and, as one line, to stderr.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before semtools_amd: one libamdhip64 per process)
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ROW_BYTES = 1024        # 256 x f32 (algorithmic bytes per row)


def make_shard(rows, seed, device):
    """Config c2 generator on the device: unit-normal rows + 1 % exact duplicates + 0.1 % zero rows."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.randn(rows, 256, device=device, generator=g, dtype=torch.float32)
    x /= x.norm(dim=1, keepdim=True)
    n_dup, n_zero = rows // 100, rows // 1000
    if n_dup:
        dst = torch.randperm(rows, device=device, generator=g)[:n_dup]
        src = torch.randint(0, rows, (n_dup,), device=device, generator=g)
        x[dst] = x[src]
    if n_zero:
        x[torch.randperm(rows, device=device, generator=g)[:n_zero]] = 0.0
    return x.contiguous()


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--settle-steps", type=int, default=256,
                    help="untimed steps run BEFORE the warmup (same count on every rank): the first ~15 ms of sustained "
                         "HBM load after idle run up to 25 %% slower (clock / power ramp, see DESIGN.md section 8)")
    ap.add_argument("--rows", type=int, default=1_000_000, help="corpus rows per GPU (c2: 1M)")
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--corpus-copies", type=int, default=2, help="distinct shards of --rows rows scanned in rotation")
    ap.add_argument("--event-every", type=int, default=8,
                    help="HIP events bracket every N-th K2 launch of the timed region (1 = every launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the c3 batched-query measurement")
    ap.add_argument("--c3-rows", type=int, default=10_000_000)
    ap.add_argument("--c3-queries", type=int, default=1000)
    ap.add_argument("--no-embed", action="store_true", help="skip the K1 (embed) measurement")
    ap.add_argument("--embed-lines", type=int, default=2_000_000)
    ap.add_argument("--no-ivfpq", action="store_true", help="skip the c5 (IVF-PQ, one GPU) measurement")
    ap.add_argument("--c5-rows", type=int, default=10_000_000)
    ap.add_argument("--c5-timeout", type=float, default=120.0, help="several ranks: seconds after which a sharded c5 leg that hangs is abandoned")
    ap.add_argument("--c5-rows-total", type=int, default=100_000_000, help="N > 1: TOTAL rows of the sharded c5 leg (split over the GPUs; 0 = skip)")
    ap.add_argument("--c5-full-rows", type=int, default=100_000_000, help="the c5 leg once more at BASELINE's named size on this one GPU (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-c4", action="store_true", help="skip BASELINE config c4 (100M rows over the N GPUs)")
    ap.add_argument("--c4-rows", type=int, default=100_000_000, help="TOTAL rows of config c4 (split over the GPUs)")
    ap.add_argument("--c4-steps", type=int, default=40)
    ap.add_argument("--c4-timeout", type=float, default=300.0, help="several ranks: seconds after which a c4 leg that hangs is abandoned")
    ap.add_argument("--no-group-issue", action="store_true", help="skip the host-issue cost of an 8-shard logical group")
    ap.add_argument("--no-small-calls", action="store_true", help="skip the small-call latency leg (c1-sized host-form searches)")
    ap.add_argument("--no-workspace", action="store_true", help="skip the workspace-mode leg (range-filtered searches, A10)")
    ap.add_argument("--ws-rows", type=int, default=10_000_000)
    ap.add_argument("--no-ingest", action="store_true", help="skip the ingest leg (tokenise || H2D || K1 through the host layer)")
    ap.add_argument("--ingest-lines", type=int, default=1_000_000)
    ap.add_argument("--detail-out", default=None, help="where the full per-leg JSON goes (default gpurun_out/bench_detail.json)")
    ap.add_argument("--min-bracketed", type=int, default=32,
                    help="at least this many K2 launches are bracketed by HIP events whatever --steps is")
    ap.add_argument("--ranks-per-process", type=int, default=0, choices=[0, 1],
                    help="0 (default when WORLD_SIZE is unset): ONE process drives all --gpus devices through one smt_group "
                         "(ncclCommInitAll); 1: one rank per process -- the command re-executes itself under torch.distributed.run")
    ap.add_argument("--single-process", action="store_true",
                    help="take the one-process GROUP path even for --gpus 1 (smt_group_create on one device: what a 1-GPU box can run of it)")
    ap.add_argument("--logical-shards", type=int, default=0,
                    help="test hook: the one-process group path over N LOGICAL ranks of cuda:0 (a 1-GPU box runs the N-shard bench code; "
                         "n_gpus stays 1 and the line says so)")
    ap.add_argument("--group-transport", default=None, choices=["peer", "rccl", "copy"],
                    help="how the per-shard k-lists meet in the one-process group (default: the library's -- peer reads)")
    return ap.parse_args(argv)


class EnvironmentProblem(Exception):
    """The run cannot start for a reason outside the code (devices missing, communicator refused): reported, exit 0."""


def error_line(args, message, visible):
    """The contract keys with value null: what a scaling record needs when the run could not start."""
    return {"metric": "chunk-vectors scanned/sec (whole job)", "value": None, "unit": "rows/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "error": str(message)[:600], "n_gpus_visible": visible,
            "config": {"workload": "c2: 1 query x 1M chunks (D=256, f32) per GPU, brute-force cosine + top-k"}}


def launcher_argv(args, argv, port=None):
    """--ranks-per-process 1 without WORLD_SIZE: the torch.distributed.run command this process becomes (pinned by a CPU test)."""
    rest = [a for a in argv]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or os.environ.get("MASTER_PORT", "29511")), os.path.abspath(__file__)] + rest


def visible_gpus():
    try:
        return int(torch.cuda.device_count()) if torch.cuda.is_available() else 0
    except Exception:
        return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    under_launcher = "WORLD_SIZE" in os.environ
    rank_env = int(os.environ.get("RANK", "0"))
    try:
        n_vis = visible_gpus()
        need = 1 if args.logical_shards else args.gpus
        if not under_launcher and n_vis < need:
            raise EnvironmentProblem(f"--gpus {args.gpus} but {n_vis} HIP device(s) visible")
        if under_launcher and int(os.environ["WORLD_SIZE"]) != args.gpus:
            raise EnvironmentProblem(f"--gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: launch with --nproc-per-node {args.gpus}, "
                                     "or without torch.distributed.run (one process then drives every GPU)")
        if not under_launcher and args.ranks_per_process == 1 and args.gpus > 1:
            cmd = launcher_argv(args, argv)
            sys.stderr.write("[bench] one rank per process: " + " ".join(cmd) + "\n")
            sys.stderr.flush()
            os.execv(cmd[0], cmd)
        run(args, under_launcher)
    except EnvironmentProblem as exc:
        if rank_env == 0:
            sys.stderr.write(f"[bench] cannot start: {exc}\n")
            sys.stdout.write(json.dumps(error_line(args, exc, visible_gpus())) + "\n")
            sys.stdout.flush()
        return 0
    except Exception as exc:   # a failure past start-up: the traceback on stderr AND a parseable line saying so (exit 1: not a result)
        import traceback
        traceback.print_exc()
        if rank_env == 0:
            sys.stdout.write(json.dumps(error_line(args, f"failed after start-up: {exc!r}", visible_gpus())) + "\n")
            sys.stdout.flush()
        return 1
    return 0


class ClockSampler:
    """Shader / memory clock and socket power of one GPU, sampled WHILE a leg runs (the same binary measures
    5-10 % apart on different leases and the record had no clock beside it).  Reads the amdgpu sysfs files every 20 ms on a thread
    (pp_dpm_sclk / pp_dpm_mclk: the line marked '*'; hwmon power1_average / power1_input in microwatts); falls back to one rocm-smi
    call after the leg when sysfs is not there.  with sampler.leg("c3"): ... ; sampler.report -> {leg: {sclk_mhz_median, ...}}."""

    def __init__(self, index=0):
        import glob
        self.report = {}
        self.dev = None
        # the sysfs directory of THIS device: by PCI address (a lease sees one GPU of a node whose sysfs lists all eight -- the first
        # closing run of round 5 read card0, an idle neighbour: 94 MHz, 243 W under every leg)
        try:
            pr = torch.cuda.get_device_properties(index)
            pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            if os.path.exists(f"/sys/bus/pci/devices/{pci}/pp_dpm_sclk"):
                self.dev = f"/sys/bus/pci/devices/{pci}"
                self.pci = pci
        except Exception:
            pass
        if self.dev is None:
            cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
            if len(cards) == 1:          # (only trusted when there is nothing to confuse it with)
                self.dev = os.path.dirname(cards[0])
        hw = sorted(glob.glob(os.path.join(self.dev, "hwmon", "hwmon*"))) if self.dev else []
        self.power_file = next((os.path.join(h, f) for h in hw for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, f))), None)

    @staticmethod
    def _current(path):
        try:
            for line in open(path):
                if "*" in line:
                    return float("".join(ch for ch in line.split(":")[1] if ch.isdigit() or ch == "."))
        except Exception:
            pass
        return None

    def _sample(self):
        s = {"sclk": self._current(os.path.join(self.dev, "pp_dpm_sclk")), "mclk": self._current(os.path.join(self.dev, "pp_dpm_mclk"))}
        try:
            s["w"] = float(open(self.power_file).read()) / 1e6 if self.power_file else None
        except Exception:
            s["w"] = None
        return s

    def leg(self, name):
        import contextlib
        import threading

        @contextlib.contextmanager
        def cm():
            samples, stop = [], threading.Event()

            def loop():
                while not stop.is_set():
                    samples.append(self._sample())
                    stop.wait(0.02)

            import gc
            gc.collect()          # (the collector is disabled while the legs run: run())
            th = None
            if self.dev:
                th = threading.Thread(target=loop, daemon=True)
                th.start()
            try:
                yield
            finally:
                stop.set()
                if th:
                    th.join(timeout=1.0)
                self.report[name] = self._summarise(samples)
        return cm()

    def _summarise(self, samples):
        if not samples:
            try:   # no sysfs: one reading right after the leg (the clocks have begun to fall by then)
                import subprocess
                d = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout)
                card = d[sorted(d)[0]]
                return {"source": "rocm-smi after the leg", **{k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "power"))}}
            except Exception as exc:
                return {"error": repr(exc)}
        out = {"source": f"amdgpu sysfs ({getattr(self, 'pci', os.path.basename(self.dev))}), 20 ms period, during the leg", "samples": len(samples)}
        for key, label in (("sclk", "sclk_mhz"), ("mclk", "mclk_mhz"), ("w", "power_w")):
            v = sorted(x[key] for x in samples if x.get(key) is not None)
            if v:
                out[label + "_median"], out[label + "_min"], out[label + "_max"] = v[len(v) // 2], v[0], v[-1]
        # a leg also generates and checks its data (idle and boost clocks in between): the clock AT THE HIGHEST POWER SAMPLE is the
        # one under its heaviest kernels
        loaded = [x for x in samples if x.get("w") is not None and x.get("sclk") is not None]
        if loaded:
            out["sclk_mhz_at_max_power"] = max(loaded, key=lambda x: x["w"])["sclk"]
        return out


def run(args, under_launcher):
    # Python's cyclic collector stays out of the timed regions: with torch imported a full collection takes 30-45 ms and is triggered
    # by allocation COUNTS -- tools/sweep_routes.py caught one inside a 20-call loop of 60 us calls (the ~39th search of a process,
    # every time).  Everything timed here is reference-counted; a collection runs at the start of every leg instead.
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Process model.  sp: ONE process drives `n_dev` devices (or logical ranks) through one smt_group; otherwise one rank per process
    # (WORLD_SIZE of them) or, at N = 1 without --single-process, a plain context with nothing to exchange.
    sp = (not under_launcher) and (args.gpus > 1 or args.single_process or args.logical_shards > 0)
    n_dev = (args.logical_shards or args.gpus) if sp else 1
    world = int(os.environ["WORLD_SIZE"]) if under_launcher else (args.gpus if not args.logical_shards else 1)
    n_shards = n_dev if sp else world          # row shards of the whole job
    # SEMTOOLS_BENCH_FORCE_EXCHANGE=1 drives the rank-per-process code path (RCCL all-gather + device merge) on ONE rank.
    # Never set by the driver; the JSON line says when it is on.
    exchange = sp or world > 1 or os.environ.get("SEMTOOLS_BENCH_FORCE_EXCHANGE") == "1"
    use_dist = exchange and not sp             # torch.distributed only joins processes
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        except Exception as exc:
            raise EnvironmentProblem(f"torch.distributed (nccl) could not start on rank {rank}/{world}: {exc!r}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # local device i of this process: cuda:i in a one-process group of real GPUs, cuda:0 for logical ranks, cuda:LOCAL_RANK otherwise
    devs = [torch.device("cuda", 0 if args.logical_shards else i) for i in range(n_dev)] if sp else [device]

    import semtools_amd as smt

    k = args.top_k
    rows = args.rows
    # a 1 GB shard is 4x the 256 MiB Infinity Cache, but rotate >= 2 copies anyway so that no residue
    # of the previous pass can flatter the HBM figure (measured effect: 149.3 us with one copy, 149.9 with two)
    n_copies = max(1, args.corpus_copies)
    shards_of = [[make_shard(rows, seed=3 + (rank if not sp else i) + 1000 * c, device=devs[i]) for c in range(n_copies)]
                 for i in range(len(devs))]       # [local device][copy]
    shards = shards_of[0]
    shard = shards[0]
    n_queries = 16
    gq = torch.Generator(device=device)
    gq.manual_seed(4)
    queries = torch.randn(n_queries, 256, device=device, generator=gq)
    queries /= queries.norm(dim=1, keepdim=True)
    queries_on = [queries if d == device else queries.to(d) for d in devs]   # the query is resident on every device

    # the library enqueues on torch's current stream, so its kernels, the RCCL
    # all-gather and the D2H copies are ordered without extra synchronisation
    # A dedicated (non-null) stream: the legacy default stream serialises against RCCL's stream, which costs
    # the one-step pipelining of the exchange its overlap (measured: 200 -> 189 us/step on one forced rank).
    if os.environ.get("SEMTOOLS_BENCH_STREAM", "side") == "side" and not sp:
        torch.cuda.synchronize(device)                        # inputs above were generated on the default stream
        torch.cuda.set_stream(torch.cuda.Stream(device))
    stream = torch.cuda.current_stream(device)
    group = None
    ginfo = None
    if exchange:
        # N > 1: the ranks join ONE library group; scan, select, exchange and merge are all enqueued by
        # smt_sharded_search_topk_device on the library's own streams (main: scans only; aux: the rest)
        from semtools_amd import dist as sdist

        for d in set(devs):
            torch.cuda.synchronize(d)
        try:
            if sp and args.logical_shards:
                group = smt.Group.logical(0, n_dev)
            elif sp:
                group = smt.Group(list(range(n_dev)))              # smt_group_create: ncclCommInitAll + peer access
            else:
                group = sdist.group_from_torch(local_rank)         # smt_group_create_rank (ncclCommInitRank), one rank included
            if args.group_transport:
                group.set_transport(args.group_transport)
        except Exception as exc:
            raise EnvironmentProblem(f"the group of {n_shards} rank(s) could not be created: {exc}")
        ctx = group.ctx(0)
        ginfo = group.info()
        ginfo["transport"] = group.transport
        ginfo["mode"] = ("one process, logical ranks on one device" if args.logical_shards else
                         "one process drives every GPU (smt_group_create: ncclCommInitAll)" if sp else
                         "one rank per process (smt_group_create_rank: ncclCommInitRank)")
        assert ginfo["n_ranks"] == n_shards and (args.logical_shards or ginfo["rccl_ranks"] == n_shards), ginfo
        corpora = [smt.ShardedCorpus(group, device_ptrs=[shards_of[i][c].data_ptr() for i in range(len(devs))],
                                     shard_rows=[rows] * len(devs)) for c in range(n_copies)]
    else:
        ctx = smt.Context(local_rank, stream=stream.cuda_stream)
        corpora = [smt.Corpus(ctx, device_ptr=sh.data_ptr(), rows=rows) for sh in shards]
    row_base = rank * rows

    # rows and distances of one query share one 2k x 8 B buffer: one D2H store per step
    ring = 64
    host = torch.empty((ring, 2, k), dtype=torch.int64).pin_memory()
    host_rows = host[:, 0]
    host_dist = host[:, 1].view(torch.float64)
    async_select = os.environ.get("SEMTOOLS_BENCH_ASYNC_SELECT", "1") != "0"
    # the verdict of EVERY timed answer (smt_search_topk_device_ex: 0 = proved the exact top-k), one pinned word per step, written
    # by the select (or, sharded: the worst over the shards, by the merging device) in stream order with the answer
    n_status = max(ring, args.steps)
    status_host = torch.full((n_status,), 7, dtype=torch.int32).pin_memory()
    status_ptr0 = status_host.data_ptr()
    want_verdicts = os.environ.get("SEMTOOLS_BENCH_NO_VERDICTS") != "1"     # (A/B hook: the plain entry points, no status words)
    if sp:
        # one process, n_dev devices: the raw C call with its pointer arrays made once (the ctypes marshalling of the Python wrapper
        # would be ~10 us of the caller's thread per step).  ONE answer per search, delivered by local device 0 into pinned memory.
        import ctypes as C

        from semtools_amd import _lib as L
        q_arrays = [(C.c_void_p * n_dev)(*[C.c_void_p(queries_on[i][j].data_ptr()) for i in range(n_dev)]) for j in range(n_queries)]
        o_arrays = [(C.c_void_p * n_dev)(*([C.c_void_p(host[r].data_ptr())] + [C.c_void_p(None)] * (n_dev - 1))) for r in range(ring)]
        s_arrays = [(C.c_void_p * n_dev)(*([C.c_void_p(status_ptr0 + 4 * j)] + [C.c_void_p(None)] * (n_dev - 1))) for j in range(n_status)]
        sharded_fn = L.lib().smt_sharded_search_topk_device_ex

    def step(i, corpora=None, slot_of=None):
        cs = corpora if corpora is not None else CORPORA
        if sp:
            rc = sharded_fn(cs[i % len(cs)]._h, q_arrays[i % n_queries], 1, k, o_arrays[i % ring], s_arrays[i % n_status] if want_verdicts else None)
            if rc:
                L.check(rc)
            return
        q = queries[i % n_queries]
        slot = host[i % ring]      # pinned host memory is device-addressable: zero-copy result delivery
        if not exchange:
            cs[i % len(cs)].search_topk_device(q.data_ptr(), 1, k, row_base, slot[0].data_ptr(), slot[1].data_ptr(),
                                               out_status_ptr=(status_ptr0 + 4 * (i % n_status)) if want_verdicts else None)
        else:
            # one call = scan (main stream) -> select -> all-gather -> merge (aux stream, overlapping the next scan)
            cs[i % len(cs)].search_topk_device([q.data_ptr()], 1, k, [slot.data_ptr()], [status_ptr0 + 4 * (i % n_status)] if want_verdicts else None)

    def sync():
        if exchange:
            group.synchronize()
            if use_dist and world > 1:
                dist.barrier()
                torch.cuda.synchronize(device)
        else:
            torch.cuda.synchronize(device)

    CORPORA = corpora
    # what a failed scaling run needs in its tail: who is here, with how much memory, before anything is timed
    try:
        free_b, total_b = torch.cuda.mem_get_info(device)
        sys.stderr.write(f"[bench rank {rank}/{world}] device {local_rank} {torch.cuda.get_device_name(device)} rows_per_gpu={rows} "
                         f"copies={len(shards)} free_hbm_gb={free_b / 1e9:.1f}/{total_b / 1e9:.1f} exchange={bool(exchange)} "
                         f"rccl_ranks={(ginfo['rccl_ranks'] if exchange else 0)} rccl_version={(ginfo.get('rccl_version') if exchange else None)}\n")
        sys.stderr.flush()
    except Exception as exc:  # diagnostics never take the run down
        sys.stderr.write(f"[bench rank {rank}] diagnostics failed: {exc!r}\n")
    local_ctxs = [group.ctx(i) for i in range(n_dev)] if sp else [ctx]

    def tune_all(key, value):
        for c in local_ctxs:
            c.set_tuning(key, value)

    def uncertain_all():
        return sum(c.uncertain_count() for c in local_ctxs)

    # The select stage of query i runs on the library's aux stream WHILE query i+1 scans (device-scope flags between
    # the two kernels); with the exchange the all-gather and the merge follow it on that stream.
    # Every step's result lands inside the timed region (sync() drains the pipeline).
    tune_all("async_select", 1 if async_select else 0)
    for i in range(args.settle_steps):
        step(i)
    sync()
    for i in range(args.warmup):
        step(i)
    sync()
    # HIP events bracket every launch of the dominant kernel (K2 scan) inside the timed region; the select
    # stage is timed in a short extra loop afterwards (each event pair costs ~5 us of stream time)
    # (every local context: at N > 1 the line carries each rank's scan / select time and the merging device's wait for the others)
    for c in local_ctxs:
        c.set_tuning("prof_select", 0)
        c.set_tuning("prof_every", args.event_every)   # an event pair costs ~6 us of stream time: sample the launches
        c.prof_enable(True)
        c.prof_reset()
    sync()
    uncertain_all()                               # reset the "exactness certificate failed" counters
    status_host.fill_(7)
    clocks = ClockSampler(local_rank)
    with clocks.leg("c2"):
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        issued = time.perf_counter() - t0             # host time to enqueue everything (diagnostic only)
        sync()
        elapsed = time.perf_counter() - t0
    n_scan, scan_ms = ctx.prof_read("scan")
    verdicts = status_host[:min(args.steps, n_status)].numpy().copy()   # one per timed step
    rank_elapsed = elapsed                                          # this rank's own clock (the MAX over ranks is taken below)
    got_rows = host_rows[(args.steps - 1) % ring].numpy().copy()   # the last timed step's answer (checked below)
    got_dist = host_dist[(args.steps - 1) % ring].numpy().copy()
    # the roofline figure must not rest on a handful of samples when the caller passes a small --steps: top the
    # bracketed launches up to --min-bracketed with more steps of the same pipeline (outside the timed region)
    extra_steps = 0
    while n_scan < args.min_bracketed and extra_steps < 64 * args.event_every:
        base_i = args.steps + extra_steps
        for i in range(base_i, base_i + args.event_every * (args.min_bracketed - n_scan)):
            step(i)
            extra_steps += 1
        sync()
        n_scan, scan_ms = ctx.prof_read("scan")
    uncertain = uncertain_all()

    def avg_us(c, name):
        n, ms = c.prof_read(name)
        return (ms / n * 1e3) if n else None

    # per LOCAL rank: its scan launches, and (the devices that take the answer) how long the merge waited for the other ranks'
    # lists once its own was ready + the merge kernel itself
    per_rank = [{"scan_avg_us": avg_us(c, "scan"), "exchange_wait_us": avg_us(c, "exchange") if exchange else None,
                 "merge_us": avg_us(c, "merge") if exchange else None} for c in local_ctxs]
    tune_all("async_select", 0)                   # the select stage is timed on its own, back to back with the scan
    for c in local_ctxs:
        c.set_tuning("prof_every", 1)
        c.set_tuning("prof_select", 1)
        c.prof_reset()
    for i in range(20):
        step(i)
    sync()
    n_sel, sel_ms = ctx.prof_read("select")
    for c, pr in zip(local_ctxs, per_rank):
        pr["select_avg_us"] = avg_us(c, "select")
        c.prof_enable(False)
        c.set_tuning("prof_select", 0)
    for pr in per_rank:
        pr["elapsed_ms"] = rank_elapsed * 1e3
        pr["rows_per_s"] = rows * args.steps / rank_elapsed
    if use_dist and world > 1:                    # one rank per process: every rank's figures travel to rank 0
        box = [None] * world
        dist.all_gather_object(box, per_rank[0])
        per_rank = box

    if use_dist:                                  # MAX over the ranks (one process: there is one clock)
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- sanity of the last step's result against an independent fp64 torch reference
    last = (args.steps - 1) % n_queries
    last_copy = (args.steps - 1) % n_copies          # the copy the last timed step scanned
    shard = shards[last_copy]
    if sp:
        # every local shard contributes its fp64 top-k (global rows = shard index x rows + local row); the merged truth must
        # equal the pipeline's output, rows included
        cand_v, cand_r = [], []
        for i, d in enumerate(devs):
            ref = 1.0 - (shards_of[i][last_copy].double() @ queries_on[i][last].double())
            lv, li = torch.topk(ref, k, largest=False)
            cand_v.append(lv.cpu())
            cand_r.append(li.cpu() + i * rows)
            del ref
        allv, allr = torch.cat(cand_v).numpy(), torch.cat(cand_r).numpy()
        order = np.lexsort((allr, allv))[:k]
        torch_ok = bool(np.allclose(got_dist, allv[order], rtol=0, atol=1e-6)) and \
            bool(np.array_equal(got_rows[np.argsort(got_dist, kind="stable")], got_rows))
        rows_ok = got_rows.tolist() == allr[order].tolist()
    else:
        ref = 1.0 - (shard.double() @ queries[last].double())
        lv, li = torch.topk(ref, k, largest=False)
        rows_ok = None
        if not exchange:
            torch_ok = bool(np.allclose(np.sort(got_dist), np.sort(lv.cpu().numpy()), rtol=0, atol=1e-6))
        else:
            # N>1: every rank contributes its local fp64 top-k; the merged truth must equal the pipeline's output
            try:
                allv = [torch.empty_like(lv) for _ in range(world)]
                dist.all_gather(allv, lv.contiguous())
                truth = torch.sort(torch.cat(allv))[0][:k].cpu().numpy()
                torch_ok = bool(np.allclose(got_dist, truth, rtol=0, atol=1e-6))
            except Exception:
                torch_ok = None

    result = {
        "metric": "chunk-vectors scanned/sec (whole job)",
        "value": n_shards * rows * args.steps / elapsed,
        "unit": "rows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "settle_steps": args.settle_steps,
        "ms_per_step": elapsed / args.steps * 1e3,
        "host_issue_ms_per_step": issued / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "c2: 1 query x 1M chunks (D=256, f32) per GPU, brute-force cosine + top-k",
                   "rows_per_gpu": rows, "dim": 256, "top_k": k, "queries_rotated": n_queries,
                   "corpus_copies_rotated": len(shards),
                   "sharding": (f"row-sharded over one smt_group of {n_shards} ranks; the per-shard k-lists meet through the group's "
                                f"'{ginfo['transport']}' transport (peer: the merge kernel reads them in place; rccl: one ncclAllGather) "
                                "+ device merge on the aux stream, overlapping the next scan") if exchange else "single shard",
                   "select_stage": "overlapped with the next query's scan (aux stream)" if async_select else "in stream order"},
    }
    if exchange and world == 1 and not sp:
        result["config"]["forced_exchange_on_one_rank"] = True
    if args.logical_shards:
        result["config"]["logical_shards_on_one_gpu"] = n_dev   # test hook: n_gpus is 1, the N shards' GPU work serialises
    if rank == 0:
        scan_us = scan_ms / max(n_scan, 1) * 1e3
        rank_scan = [pr["scan_avg_us"] for pr in per_rank if pr.get("scan_avg_us")]
        if len(per_rank) > 1 and len(rank_scan) == len(per_rank):
            scan_us = max(rank_scan)              # the roofline figure of a sharded job is its SLOWEST rank's (min frac over ranks)
        achieved = rows * ROW_BYTES / (scan_us * 1e-6) / 1e9 if n_scan else None
        traffic, traffic_source = measured_traffic("c2", rows)
        result["roofline"] = {
            "kernel": "scan_topk_kernel (K2)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
            "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": rows * ROW_BYTES, "avg_kernel_us": scan_us, "launches": n_scan,
            "launches_note": f"HIP events on every {args.event_every}-th launch of the {args.steps} timed steps"
                             + (f" + {extra_steps} more steps of the same pipeline (--min-bracketed)" if extra_steps else ""),
            "select_avg_us": sel_ms / max(n_sel, 1) * 1e3,
        }
        if len(per_rank) > 1:
            result["roofline"]["frac_is"] = "the minimum over the ranks (slowest rank's average scan launch)"
            result["roofline"]["frac_per_rank"] = [rows * ROW_BYTES / (u * 1e-6) / 1e9 / HBM_PEAK_GBPS if u else None
                                                   for u in (pr.get("scan_avg_us") for pr in per_rank)]
        if exchange:
            # what a SCALE record needs to be explained: who was slow (scan / select per rank), what the exchange cost the merging
            # device (wait for the slowest list + merge), which transport really ran, every rank's own clock
            merging = per_rank[0]
            result["ranks"] = {
                "process_model": "one process, logical ranks" if args.logical_shards else "one process" if sp else "one rank per process",
                "transport": ginfo["transport"], "rccl_ranks": ginfo["rccl_ranks"], "n_ranks": n_shards,
                "scan_avg_us": [pr.get("scan_avg_us") for pr in per_rank], "select_avg_us": [pr.get("select_avg_us") for pr in per_rank],
                "exchange_wait_us": merging.get("exchange_wait_us"), "merge_us": merging.get("merge_us"),
                "exchange_wait_us_per_rank": [pr.get("exchange_wait_us") for pr in per_rank],
                "rank_rows_per_s": [pr.get("rows_per_s") for pr in per_rank], "rank_elapsed_ms": [pr.get("elapsed_ms") for pr in per_rank],
                "note": "scan / select: HIP events around each rank's own launches; exchange_wait: from the merging device's own k-list being "
                        "ready to every rank's list being there (rank skew + transport); merge: the merge kernel.  One-process groups: rank 0 "
                        "takes the answer, so only it has exchange / merge figures; rank_elapsed is the one caller's clock for all of them",
            }
        result["checks"] = {"torch_fp64_topk_distances_match": torch_ok,
                            "selects_without_exactness_certificate": uncertain,
                            "every_timed_answer_proved_exact": f"{int((verdicts == 0).sum())}/{len(verdicts)}" if want_verdicts else "not asked/asked"}
        if rows_ok is not None:
            result["checks"]["rows_match_fp64_topk_over_all_shards"] = rows_ok
        if exchange:
            result["config"]["group"] = ginfo

    # the legs below describe ONE GPU driven through a plain context on torch's stream: they run at N = 1 only, and not in the
    # one-process group mode (whose contexts own their streams)
    solo = world == 1 and not sp
    if sp:
        # the c2 shards of the other devices are not needed any more (c4 wants the memory)
        for c in corpora:
            c.close()
        shards_of = [shards_of[0]]
        torch.cuda.empty_cache()
    # (K1 before the legs that allocate and free 10-100 GB: the 4 GB table of its uniform-id case is gathered ~5 % slower when it is
    # allocated after such a cycle in the same process -- 6.2 ms alone, 6.55-6.9 ms after c4, same box, same binary)
    if rank == 0 and solo and not args.no_embed:
        try:
            with clocks.leg("embed"):
                result["embed"] = bench_embed(smt, ctx, device, args.embed_lines)
        except Exception as exc:
            result["embed"] = {"error": repr(exc)}

    if not args.no_c4:
        # With several ranks c4 is collective: a rank that fails inside it (out of memory, a communicator error) leaves the others
        # waiting in ncclAllGather for ever, and the c2 figures -- complete at this point -- would never be printed.  A watchdog on
        # every rank abandons the leg after --c4-timeout seconds: rank 0 prints the line without it, everybody leaves.
        watchdog = None
        fake_hang = os.environ.get("SEMTOOLS_BENCH_FAKE_C4_HANG") == "1"   # tests: the leg sleeps for ever, on one rank too
        if n_shards > 1 or fake_hang:
            import threading

            def abandon():
                sys.stderr.write(f"[bench rank {rank}/{world}] c4 gave no answer within {args.c4_timeout} s: leg abandoned\n")
                sys.stderr.flush()
                if rank == 0:
                    result["c4"] = {"error": f"no answer within {args.c4_timeout} s on {world} ranks: leg abandoned (the c2 figures were complete)"}
                    emit_line(result, args)
                os._exit(0)

            watchdog = threading.Timer(args.c4_timeout, abandon)
            watchdog.daemon = True
            watchdog.start()
        try:    # every rank takes part (row-sharded corpus, collective exchange); rank 0 reports
            if fake_hang:
                time.sleep(1e9)
            with clocks.leg("c4"):
                if sp:
                    c4 = bench_c4_one_process(smt, args, devs, group, local_ctxs, k, queries_on, host, world)
                else:
                    c4 = bench_c4(smt, args, device, rank, world, group, ctx, k, queries, host)
        except Exception as exc:
            c4 = {"error": repr(exc)}
        if use_dist and world > 1:
            try:    # (a rank that failed alone must not be waited for: only when every rank got here is the leg over)
                dist.barrier()
            except Exception:
                pass
        if watchdog is not None:
            watchdog.cancel()
        if rank == 0:
            result["c4"] = c4

    if n_shards > 1 and not args.no_ivfpq and args.c5_rows_total > 0:
        # c5 on the job's GPUs: collective like c4 (shared-centroid build, the exchange of every search): the same watchdog
        import threading

        def abandon_c5():
            sys.stderr.write(f"[bench rank {rank}/{world}] the sharded c5 leg gave no answer within {args.c5_timeout} s: leg abandoned\n")
            sys.stderr.flush()
            if rank == 0:
                result["ivfpq_sharded"] = {"error": f"no answer within {args.c5_timeout} s on {n_shards} ranks: leg abandoned"}
                emit_line(result, args)
            os._exit(0)

        watchdog5 = threading.Timer(args.c5_timeout, abandon_c5)
        watchdog5.daemon = True
        watchdog5.start()
        try:
            with clocks.leg("ivfpq_sharded"):
                c5s = bench_c5_sharded(smt, args, devs, rank, n_shards, group, k)
        except Exception as exc:
            c5s = {"error": repr(exc)}
        if use_dist and world > 1:
            try:
                dist.barrier()
            except Exception:
                pass
        watchdog5.cancel()
        if rank == 0:
            result["ivfpq_sharded"] = c5s

    if rank == 0 and solo and not args.no_secondary:
        try:
            with clocks.leg("c3"):
                result["secondary"] = bench_c3(smt, ctx, device, args.c3_rows, args.c3_queries, k)
        except Exception as exc:  # never let an auxiliary leg take the headline line down with it
            result["secondary"] = {"error": repr(exc)}

    if rank == 0 and solo and not args.no_group_issue:
        try:
            with clocks.leg("group_issue"):
                result["group_issue"] = bench_group_issue(smt, device)
        except Exception as exc:
            result["group_issue"] = {"error": repr(exc)}

    if rank == 0 and solo and not args.no_small_calls:
        try:
            result["small_calls"] = bench_small_calls(smt, ctx, device, shard, k)
        except Exception as exc:
            result["small_calls"] = {"error": repr(exc)}

    if rank == 0 and solo and not args.no_workspace:
        try:
            with clocks.leg("workspace"):
                result["workspace"] = bench_workspace(smt, ctx, device, args.ws_rows, k)
        except Exception as exc:
            result["workspace"] = {"error": repr(exc)}

    if rank == 0 and solo and not args.no_ingest:
        try:
            result["ingest"] = bench_ingest(smt, ctx, args.ingest_lines)
        except Exception as exc:
            result["ingest"] = {"error": repr(exc)}

    if rank == 0 and solo and not args.no_ivfpq:
        try:
            with clocks.leg("ivfpq"):
                result["ivfpq"] = bench_c5(smt, ctx, device, args.c5_rows, k)
        except Exception as exc:  # the approximate index is a "next" row: never let it break the headline line
            result["ivfpq"] = {"error": repr(exc)}

    if rank == 0 and solo and not args.no_ivfpq and args.c5_full_rows > 0:
        try:
            with clocks.leg("ivfpq_full"):
                result["ivfpq_full"] = bench_c5_full(smt, ctx, device, args.c5_full_rows, k)
        except Exception as exc:
            result["ivfpq_full"] = {"error": repr(exc)}

    if rank == 0 and solo and not args.no_cpu_baseline:
        try:
            from oracle import oracle as orc

            host_np = shard.cpu().numpy()
            hq = queries.cpu().numpy()
            # parity of the last measured step against the oracle (indices exact, distances 1e-5 / f64-exact)
            res = orc.search_documents(host_np, [rows], hq[last], n_lines=0, top_k=k, accurate=True)
            result["checks"]["oracle_rows_match"] = [r["match_line"] for r in res] == got_rows.tolist()
            result["checks"]["oracle_dist_max_abs_diff"] = float(np.abs(np.array([r["distance"] for r in res]) - got_dist).max())
            # (i) reference-faithful SCALAR port: single thread, every row's result materialised, stable sort, take(k)
            t_cpu, n_cpu = 0.0, 0
            while t_cpu < args.cpu_seconds / 3 and n_cpu < 64:
                c0 = time.perf_counter()
                orc.search_documents(host_np, [rows], hq[n_cpu % n_queries], n_lines=3, top_k=k, accurate=False)
                t_cpu += time.perf_counter() - c0
                n_cpu += 1
            # (ii) the same control flow with the cosine the reference really runs on this host: simsimd dispatches to
            # its AVX-512 / AVX2 f32 kernel at run time (oracle/cpu_fast.c: orc_search_documents_simd).  Single thread,
            # like the reference.  THIS is the stated baseline.
            simd_res = orc.search_documents_simd(host_np, hq[last], n_lines=3, top_k=k)
            result["checks"]["simd_port_rows_match"] = [r["match_line"] for r in simd_res] == got_rows.tolist()
            t_simd, n_simd = 0.0, 0
            while t_simd < args.cpu_seconds and n_simd < 256:
                c0 = time.perf_counter()
                orc.search_documents_simd(host_np, hq[n_simd % n_queries], n_lines=3, top_k=k)
                t_simd += time.perf_counter() - c0
                n_simd += 1
            # (iii) "fair CPU" variant: threaded, vectorised, bounded per-thread lists.  Thread count: best of a few
            # candidates (containers often expose more logical CPUs than they may use)
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            best_t, best_rate = 1, 0.0
            for t_try in sorted({1, 8, 16, 32, 64, 128, avail} & set(range(1, avail + 1))):
                f0 = time.perf_counter()
                orc.scan_topk_threads(host_np, hq[0], k, t_try)
                rate = rows / (time.perf_counter() - f0)
                if rate > best_rate:
                    best_t, best_rate = t_try, rate
            ncores = best_t
            f0 = time.perf_counter()
            n_fair = 0
            while time.perf_counter() - f0 < min(args.cpu_seconds, 4.0):
                orc.scan_topk_threads(host_np, hq[n_fair % n_queries], k, ncores)
                n_fair += 1
            t_fair = time.perf_counter() - f0
            result["cpu_baseline"] = {
                "value": rows * n_simd / t_simd, "unit": "rows/s", "cores": 1, "kind": "port", "variant": "port-simd",
                "sample_short": f"{n_simd} queries x {rows} rows, reference control flow (cosine per row, sort all, take k), {orc.simd_backend()} f32, 1 thread",
                "sample": f"{n_simd} queries x {rows} rows (the shard of the last timed step copied back): the reference's "
                          "control flow (src/search/mod.rs:84-119: one cosine per row, a record for every row, stable sort "
                          f"of all records, take k) with a {orc.simd_backend()} f32 cosine as simsimd dispatches on this "
                          "host; gcc -O3, single thread as in the reference; String clones per record not modelled",
                "scalar_port_value": rows * n_cpu / t_cpu,
                "scalar_port_note": "same control flow, scalar no-FMA cosine (oracle/semtools_oracle.c, gcc -O2): the parity "
                                    "oracle's arithmetic, 5-10x slower per row than what the reference executes",
                "fair_threads_value": rows * n_fair / t_fair, "fair_threads_cores": ncores,
                "host_cpu": _cpu_model(),
            }
        except Exception as exc:
            result["cpu_baseline"] = {"error": repr(exc)}
    if rank == 0:
        result["clocks"] = clocks.report
    # The JSON line must be the LAST thing on stdout.  RCCL prints a version banner through C stdio when a communicator is
    # created; redirected to a file, that text sits in libc's buffer until exit and would land BEHIND the line.  Flush C
    # stdio first, then print and flush the line.
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if use_dist and world > 1:
        dist.barrier()          # every rank's banner is out before rank 0 writes the line
    if rank == 0:
        emit_line(result, args)
    if use_dist:
        dist.destroy_process_group()


def emit_line(result, args):
    """Rank 0: the full per-leg objects to the detail file and stderr, the compact line LAST on stdout."""
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    line = compact_line(result)
    try:
        path = args.detail_out or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(result, f, indent=1)
        line["detail_file"] = os.path.relpath(path, ROOT)
    except Exception:
        line["detail_file"] = None
    sys.stderr.write("[bench detail] " + json.dumps(result) + "\n")
    sys.stderr.flush()
    sys.stdout.write(json.dumps(line) + "\n")
    sys.stdout.flush()


def _g(d, *path, default=None):
    for key in path:
        if not isinstance(d, dict) or key not in d:
            return default
        d = d[key]
    return d


def _r(v, digits=4):
    """Numbers of the compact line carry 4-5 significant digits: the full precision is in the detail file."""
    if isinstance(v, bool) or v is None or isinstance(v, (str, int)):
        return v
    try:
        return float(f"{float(v):.{digits + 1}g}")
    except Exception:
        return v


def collect_checks(d, prefix=""):
    """Every check of every leg -> (n_checks, [failed names]).  bool: must be True; "a/b" strings: a == b; *_max_abs_diff: <= 1e-5
    (the distance contract); selects_without_exactness_certificate: 0 (a device-resident leg is not re-answered by the host)."""
    n, failed = 0, []
    if not isinstance(d, dict):
        return n, failed
    for key, v in d.items():
        name = f"{prefix}{key}"
        if key == "checks" and isinstance(v, dict):
            for ck, cv in v.items():
                n += 1
                ok = True
                if isinstance(cv, bool):
                    ok = cv
                elif cv is None:
                    ok = False
                elif isinstance(cv, str) and "/" in cv:
                    a, _, b = cv.partition("/")
                    ok = a.strip() == b.strip()
                elif ck.endswith("max_abs_diff"):
                    ok = float(cv) <= 1e-5
                elif ck == "selects_without_exactness_certificate":
                    ok = int(cv) == 0
                if not ok:
                    failed.append(f"{prefix}{ck}={cv}")
        elif isinstance(v, dict):
            if "error" in v and len(v) == 1:
                n += 1
                failed.append(f"{name}: {str(v['error'])[:80]}")
            else:
                n2, f2 = collect_checks(v, name + ".")
                n += n2
                failed += f2
    return n, failed


def compact_line(d):
    """The one stdout line: contract keys + c2 roofline + cpu_baseline (short strings) + every leg's numbers as flat keys."""
    line = {key: d.get(key) for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                        "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _r(line["value"], 6), _r(line["ms_per_step"], 5)
    cfg = d.get("config", {})
    line["config"] = {"workload": cfg.get("workload"), "rows_per_gpu": cfg.get("rows_per_gpu"), "dim": cfg.get("dim"), "top_k": cfg.get("top_k"),
                      "corpus_copies_rotated": cfg.get("corpus_copies_rotated"),
                      "sharding": "row-sharded over one smt_group, k-lists exchanged inside the library + device merge" if cfg.get("group") else "single shard"}
    if cfg.get("group"):
        line["config"]["group"] = {key: _g(cfg, "group", key) for key in ("n_ranks", "n_local", "rccl_ranks", "rccl_version", "transport", "mode")}
        line["config"]["parallelism"] = f"row shards x {_g(cfg, 'group', 'n_ranks')}"
    if cfg.get("logical_shards_on_one_gpu"):
        line["config"]["logical_shards_on_one_gpu"] = cfg["logical_shards_on_one_gpu"]
    if cfg.get("forced_exchange_on_one_rank"):
        line["config"]["forced_exchange_on_one_rank"] = True
    rf = d.get("roofline")
    if rf:
        line["roofline"] = {"kernel": rf.get("kernel"), "bound": rf.get("bound"), "achieved": _r(rf.get("achieved")), "peak": rf.get("peak"),
                            "unit": rf.get("unit"), "frac": _r(rf.get("frac")), "traffic": rf.get("traffic"),
                            "algorithmic_bytes_per_launch": rf.get("algorithmic_bytes_per_launch"), "avg_kernel_us": _r(rf.get("avg_kernel_us")),
                            "launches": rf.get("launches"), "select_avg_us": _r(rf.get("select_avg_us")),
                            "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE x2, separate run)"}
    line["host_issue_ms_per_step"] = _r(d.get("host_issue_ms_per_step"))
    rk = d.get("ranks")
    if rk:     # N > 1 (or a forced / logical exchange): per-rank figures, 3 significant digits
        def arr(key):
            return [_r(v, 3) for v in rk.get(key) or []]
        line["ranks"] = {"process_model": rk.get("process_model"), "transport": rk.get("transport"), "rccl_ranks": rk.get("rccl_ranks"),
                         "scan_avg_us": arr("scan_avg_us"), "select_avg_us": arr("select_avg_us"),
                         "exchange_wait_us": _r(rk.get("exchange_wait_us"), 3), "merge_us": _r(rk.get("merge_us"), 3),
                         "rank_rows_per_s": arr("rank_rows_per_s")}
        if rf and rf.get("frac_per_rank"):
            line["roofline"]["frac_is"] = "min over ranks"
    cb = d.get("cpu_baseline")
    if cb and "error" not in cb:
        line["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "variant": cb.get("variant"), "sample": cb.get("sample_short") or str(cb.get("sample"))[:118],
                                "scalar_port_value": _r(cb.get("scalar_port_value")), "fair_threads_value": _r(cb.get("fair_threads_value")),
                                "fair_threads_cores": cb.get("fair_threads_cores"), "host_cpu": cb.get("host_cpu")}
    elif cb:
        line["cpu_baseline"] = {"error": str(cb["error"])[:118]}

    def put(key, *path, digits=4):
        v = _g(d, *path)
        if v is not None:
            line[key] = _r(v, digits)

    # c3: queries/sec at a 10 M-chunk corpus (the second half of BASELINE.json's metric)
    put("c3_queries_per_s", "secondary", "value")
    put("c3_ms_per_batch", "secondary", "ms_per_batch")
    put("c3_frac_of_2p5PF", "secondary", "roofline", "frac")
    put("c3_frac_of_sustained_1p4PF", "secondary", "roofline", "sustained_peak_random_operands", "frac_of_lds_fed")
    put("c3_mode", "secondary", "roofline", "kernel")
    put("c3_image_queries_per_s", "secondary", "operand_image", "queries_per_s")
    put("c3_image_ms_per_batch", "secondary", "operand_image", "ms_per_batch")
    put("c3_image_frac_of_2p5PF", "secondary", "operand_image", "roofline", "frac")
    # MFMA-busy fraction of the main level of that batch, from counters: a SEPARATE rocprofv3 --pmc run of the same binary
    # (tools/gpu_r06_pmc.sh -> profiles/r06_k3/r06_k3_summary.json); counters cannot be collected inside a timed run
    if "c3_queries_per_s" in line:
        try:
            k3 = json.load(open(os.path.join(ROOT, "profiles", "r06_k3", "r06_k3_summary.json")))
            line["c3_mfma_busy_frac"] = k3["1000q_f32rows"]["mfma_busy_frac_main_level"]
            line["c3_image_mfma_busy_frac"] = k3["1000q_image"]["mfma_busy_frac_main_level"]
            line["c3_mfma_busy_source"] = "profiles/r06_k3/r06_k3_summary.json (rocprofv3 --pmc, separate run)"
        except Exception:
            pass
    put("c3_f32mfma_queries_per_s", "secondary", "roofline_f32_mfma", "queries_per_s")
    put("c3_f32mfma_frac_of_157TF", "secondary", "roofline_f32_mfma", "frac")
    put("c3_1q_10M_f32_scan_ms", "secondary", "single_query_same_corpus", "f32_scan_ms")
    put("c3_1q_10M_image_ms", "secondary", "single_query_same_corpus", "image_ms")
    put("c3_1q_10M_image_frac_hbm_512B", "secondary", "single_query_same_corpus", "image_frac_of_hbm_at_512B_per_row")
    # c4: 1 query x 100 M chunks over the job's GPUs
    put("c4_rows_per_s", "c4", "value", digits=5)
    put("c4_ms_per_query", "c4", "ms_per_query")
    put("c4_frac_hbm", "c4", "roofline", "frac")
    put("c4_n_gpus", "c4", "n_gpus")
    put("c4_image_ms_per_query", "c4", "operand_image", "ms_per_query")
    put("c4_image_frac_hbm_512B", "c4", "operand_image", "frac_of_hbm_at_512B_per_row")
    # workspace mode (A10: path-subset filter + score threshold + top-k)
    put("ws_rows_scanned", "workspace", "rows_scanned")
    put("ws_1q_ms", "workspace", "one_query", "ms_per_call")
    put("ws_1q_rows_per_s", "workspace", "one_query", "rows_per_s")
    put("ws_1q_frac_hbm", "workspace", "one_query", "roofline", "frac")
    put("ws_1q_image_ms", "workspace", "one_query_image", "ms_per_call")
    put("ws_1q_image_frac_hbm_512B", "workspace", "one_query_image", "frac_of_hbm_at_512B_per_row")
    put("ws_batch_queries", "workspace", "batch", "queries")
    put("ws_batch_queries_per_s", "workspace", "batch", "queries_per_s")
    put("ws_batch_ms", "workspace", "batch", "ms_per_call")
    put("ws_batch_cost_per_scanned_row_vs_unfiltered", "workspace", "batch", "cost_per_scanned_row_vs_unfiltered")
    put("ws_batch_image_queries_per_s", "workspace", "batch_image", "queries_per_s")
    put("ws_batch_image_ms", "workspace", "batch_image", "ms_per_call")
    put("ws_batch_image_cost_per_scanned_row_vs_unfiltered", "workspace", "batch_image", "cost_per_scanned_row_vs_unfiltered")
    # host-form calls where the reference lives (c1: one query over 1000 lines): us per synchronous call
    put("c1_call_us", "small_calls", "delivered", "c1_1000_rows_1q_top3_us", digits=3)
    put("c1_3q_call_us", "small_calls", "delivered", "c1_1000_rows_3q_top3_us", digits=3)
    put("one_row_call_us", "small_calls", "delivered", "one_row_1q_us", digits=3)
    put("rows_64k_call_us", "small_calls", "delivered", "rows_65536_1q_top10_us", digits=3)
    put("c2_host_call_us", "small_calls", "delivered", "c2_1M_rows_1q_us", digits=4)
    put("c1_call_us_copy_sync", "small_calls", "copy_and_synchronize", "c1_1000_rows_1q_top3_us", digits=3)
    put("c2_host_call_us_copy_sync", "small_calls", "copy_and_synchronize", "c2_1M_rows_1q_us", digits=4)
    # several shards driven by one host thread (logical group on one GPU: the issue cost, not the collective)
    put("group_issue_us_8_logical_shards", "group_issue", "host_issue_us_per_search")
    put("group_launches_only_us_8_shards", "group_issue", "one_thread_issues_every_shard_us")
    put("group_issue_copy_transport_us", "group_issue", "copy_transport_us")
    # shader clock (MHz, median of the samples taken DURING the leg) and socket power beside the legs that vary from lease to lease
    # (a leg also generates and checks its data: the clock at the HIGHEST power sample is the one under its heaviest kernels)
    for leg in ("c2", "c3", "c4", "embed"):
        put(f"clk_{leg}_mhz", "clocks", leg, "sclk_mhz_at_max_power", digits=3)
        put(f"pwr_{leg}_max_w", "clocks", leg, "power_w_max", digits=3)
    # K1 (embed) and the host step in front of it
    put("embed_lines_per_s_zipf", "embed", "zipf_ids_500k_table", "lines_per_s")
    put("embed_frac_hbm_zipf_measured_traffic", "embed", "zipf_ids_500k_table", "roofline", "frac")
    put("embed_lines_per_s_uniform", "embed", "uniform_ids_4M_table", "lines_per_s")
    put("embed_frac_hbm_uniform", "embed", "uniform_ids_4M_table", "roofline", "frac")
    put("embed_kernel_ms_uniform", "embed", "uniform_ids_4M_table", "kernel_ms")
    put("ingest_lines_per_s", "ingest", "lines_per_s")
    put("ingest_text_MB_per_s", "ingest", "text_MB_per_s")
    put("ingest_wordpiece_lines_per_s", "ingest", "wordpiece_tokenizer_json", "lines_per_s")
    put("ingest_cores", "ingest", "cores")
    # c5 on one GPU
    put("ivf_recall_at_k", "ivfpq", "recall_at_k_vs_exact")
    put("ivf_queries_per_s", "ivfpq", "queries_per_s")
    put("ivf_build_s", "ivfpq", "build_s")
    put("ivf_adc_bound", "ivfpq", "roofline", "bound")
    put("ivf_adc_frac", "ivfpq", "roofline", "frac")
    # ... sharded over the job's GPUs (N > 1)
    put("ivf_sharded_build_s", "ivfpq_sharded", "build_s")
    put("ivf_sharded_recall_at_k", "ivfpq_sharded", "recall_at_k_vs_exact_sharded_search")
    put("ivf_sharded_queries_per_s", "ivfpq_sharded", "queries_per_s")
    put("ivf_sharded_rows_total", "ivfpq_sharded", "rows_total")
    # ... and at c5's named size (100 M rows on this one GPU)
    put("ivf100m_build_s", "ivfpq_full", "build_s")
    put("ivf100m_recall_at_k", "ivfpq_full", "nprobe_8", "recall_at_k_vs_exact")
    put("ivf100m_queries_per_s", "ivfpq_full", "nprobe_8", "queries_per_s")
    put("ivf100m_np1_recall_at_k", "ivfpq_full", "nprobe_1", "recall_at_k_vs_exact")
    put("ivf100m_np1_queries_per_s", "ivfpq_full", "nprobe_1", "queries_per_s")
    put("ivf100m_np32_recall_at_k", "ivfpq_full", "nprobe_32", "recall_at_k_vs_exact")
    put("ivf100m_np32_queries_per_s", "ivfpq_full", "nprobe_32", "queries_per_s")
    put("ivf100m_np128_recall_at_k", "ivfpq_full", "nprobe_128", "recall_at_k_vs_exact")
    put("ivf100m_np128_queries_per_s", "ivfpq_full", "nprobe_128", "queries_per_s")
    put("ivf100m_pq_build_s", "ivfpq_full", "global_pq_m32", "build_s")
    for np_ in (8, 32):
        put(f"ivf100m_pq_np{np_}_recall_at_k", "ivfpq_full", "global_pq_m32", f"nprobe_{np_}", "recall_at_k_vs_exact")
        put(f"ivf100m_pq_np{np_}_queries_per_s", "ivfpq_full", "global_pq_m32", f"nprobe_{np_}", "queries_per_s")
    for coding, tag in (("per_list_pca", "ivf_hard"), ("global_pq_m32", "ivf_hard_pq")):
        for np_ in (8, 32, 128):
            put(f"{tag}_np{np_}_recall_at_k", "ivfpq", "hard_corpus", coding, f"nprobe_{np_}", "recall_at_k_vs_exact")
            put(f"{tag}_np{np_}_queries_per_s", "ivfpq", "hard_corpus", coding, f"nprobe_{np_}", "queries_per_s")
    put("ivf_pq_recall_at_k", "ivfpq", "global_pq_m32", "recall_at_k_vs_exact")
    put("ivf_pq_queries_per_s", "ivfpq", "global_pq_m32", "queries_per_s")
    put("ivf_pq_build_s", "ivfpq", "global_pq_m32", "build_s")
    n_checks, failed = collect_checks(d)
    line["checks_total"] = n_checks
    line["checks_ok"] = not failed
    line["checks_failed"] = [f[:100] for f in failed[:12]]
    return line


def measured_traffic(leg, rows=None):
    """HBM bytes per launch of a leg's dominant kernel as MEASURED with rocprofv3 --pmc FETCH_SIZE (x2 on gfx950,
    MI355X_MICROARCH.md section HBM) in an earlier profiling run, and the file under profiles/ the figure comes from --
    the counters cannot be collected inside a timed bench run.  (None, reason) when no figure matches this workload."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        entry = json.load(open(tfile)).get("legs", {}).get(leg)
    except Exception:
        return None, "profiles/traffic.json unreadable"
    if not entry:
        return None, f"no PMC figure recorded for leg '{leg}'"
    if rows is not None and entry.get("rows") != rows:
        return None, f"PMC figure recorded for {entry.get('rows')} rows, this run has {rows}"
    return entry.get("hbm_bytes_per_launch"), f"{entry.get('source')} ({entry.get('counter', 'FETCH_SIZE x2')}; not measured by this run)"


def _traffic_entry(leg):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("legs", {}).get(leg)
    except Exception:
        return None


def torch_topk_fp64(x, qv, k, chunk=2_000_000):
    """(distances, rows) of the k smallest fp64 cosine distances of the rows of x (unit rows / unit query), chunked:
    x.double() of a 100M-row shard would not fit."""
    best_v = best_i = None
    qd = qv.double()
    for b in range(0, x.shape[0], chunk):
        d = 1.0 - (x[b:b + chunk].double() @ qd)
        v, i = torch.topk(d, min(k, d.numel()), largest=False)
        i = i + b
        if best_v is not None:
            v, i = torch.cat([best_v, v]), torch.cat([best_i, i])
            v, sel = torch.topk(v, min(k, v.numel()), largest=False)
            i = i[sel]
        best_v, best_i = v, i
    if best_v is None:
        return torch.empty(0, dtype=torch.float64, device=x.device), torch.empty(0, dtype=torch.int64, device=x.device)
    return best_v, best_i


def bench_c4(smt, args, device, rank, world, group, ctx, k, queries, host):
    """BASELINE config c4: 1 query x 100M chunks, row-sharded over the N GPUs of the job (100M / N rows per GPU,
    generated on the device per shard), per-shard scan + select, ONE all-gather of the k-lists + merge.  At N = 1 it
    is the 1-GPU point of that curve: the whole 102.4 GB corpus resident on one MI355X."""
    total = args.c4_rows
    per = -(-total // world)
    my_rows = max(0, min(per, total - rank * per))
    x = torch.empty((my_rows, 256), device=device, dtype=torch.float32)
    g = torch.Generator(device=device)
    g.manual_seed(3 + rank)
    step_rows = 2_000_000
    c = None
    for b in range(0, my_rows, step_rows):
        e = min(my_rows, b + step_rows)
        c = torch.randn(e - b, 256, device=device, generator=g)
        c /= c.norm(dim=1, keepdim=True)
        x[b:e] = c
    del c
    torch.cuda.synchronize(device)
    ring = host.shape[0]
    if group is None:
        corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=my_rows)

        def one(i):
            slot = host[i % ring]
            corpus.search_topk_device(queries[i % len(queries)].data_ptr(), 1, k, 0, slot[0].data_ptr(), slot[1].data_ptr())

        def sync():
            torch.cuda.synchronize(device)
    else:
        corpus = smt.ShardedCorpus(group, device_ptrs=[x.data_ptr()], shard_rows=[my_rows])

        def one(i):
            corpus.search_topk_device([queries[i % len(queries)].data_ptr()], 1, k, [host[i % ring].data_ptr()])

        def sync():
            group.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize(device)
    ctx.set_tuning("async_select", 1)
    ctx.set_tuning("prof_select", 0)
    ctx.set_tuning("prof_every", 1)
    for i in range(3):
        one(i)
    sync()
    ctx.prof_enable(True)
    ctx.prof_reset()
    ctx.uncertain_count()
    sync()
    t0 = time.perf_counter()
    for i in range(args.c4_steps):
        one(i)
    sync()
    elapsed = time.perf_counter() - t0
    n_scan, scan_ms = ctx.prof_read("scan")
    ctx.prof_enable(False)
    ctx.set_tuning("async_select", 0)
    ctx.set_tuning("prof_select", 1)
    uncertain = ctx.uncertain_count()
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # The same queries over the shard's fp16 operand image (512 B per row; what a corpus the library owns keeps once it is
    # searched repeatedly, here requested for the adopted tensor): reported beside the headline, which stays the f32 scan.
    image_leg = None
    if group is None and world == 1:
        try:
            t0 = time.perf_counter()
            corpus.prepack()
            ctx.synchronize()
            build_s = time.perf_counter() - t0
            ir = torch.empty(1, k, dtype=torch.int64, device=device)
            idist = torch.empty(1, k, dtype=torch.float64, device=device)
            qi = queries[(args.c4_steps - 1) % len(queries)]
            corpus.search_topk_device(qi.data_ptr(), 1, k, 0, ir.data_ptr(), idist.data_ptr())
            ctx.synchronize()
            ctx.uncertain_count()
            n_img = max(4, args.c4_steps // 4)
            t0 = time.perf_counter()
            for _ in range(n_img):
                corpus.search_topk_device(qi.data_ptr(), 1, k, 0, ir.data_ptr(), idist.data_ptr())
            ctx.synchronize()
            img_s = (time.perf_counter() - t0) / n_img
            last_i = args.c4_steps - 1
            same = bool((ir[0].cpu().numpy() == host[last_i % ring, 0].numpy()).all()) and \
                bool((idist[0].cpu().numpy() == host[last_i % ring, 1].view(torch.float64).numpy()).all())
            image_leg = {"ms_per_query": img_s * 1e3, "rows_per_s": my_rows / img_s, "build_ms": build_s * 1e3,
                         "image_bytes": corpus.image_bytes, "frac_of_hbm_at_512B_per_row": my_rows * 512 / img_s / (HBM_PEAK_GBPS * 1e9),
                         "answers_identical_to_f32_scan": same, "selects_without_exactness_certificate": ctx.uncertain_count(),
                         "note": "whole calls (levels, selects, delivery; no overlap between queries); gemm_rowreg_kernel<1, true>"}
            corpus.prepack(False)
        except Exception as exc:
            image_leg = {"error": repr(exc)}
    last = args.c4_steps - 1
    got_dist = host[last % ring, 1].view(torch.float64).numpy().copy()
    got_rows = host[last % ring, 0].numpy().copy()
    lv, li = torch_topk_fp64(x, queries[last % len(queries)], k)
    li = li + rank * per                                  # global rows: shards are contiguous ranges in rank order
    if world > 1:
        pad = torch.full((k,), float("inf"), dtype=torch.float64, device=device)
        pad[: lv.numel()] = lv
        padi = torch.full((k,), -1, dtype=torch.int64, device=device)
        padi[: li.numel()] = li
        allv = [torch.empty_like(pad) for _ in range(world)]
        alli = [torch.empty_like(padi) for _ in range(world)]
        dist.all_gather(allv, pad)
        dist.all_gather(alli, padi)
        lv, order = torch.sort(torch.cat(allv))
        lv, li = lv[:k], torch.cat(alli)[order][:k]
    ok = bool(np.allclose(got_dist, lv.cpu().numpy(), rtol=0, atol=1e-6))
    rows_ok = bool(((got_rows >= 0) & (got_rows < total)).all())
    # the returned ROW INDICES against the independent fp64 top-k (ties -- none in this random corpus -- would come back in
    # ascending row order; compare as sorted (distance, row) pairs)
    want_rows = li.cpu().numpy()
    rows_match = bool(got_rows.tolist() == want_rows.tolist())
    corpus.close()
    del x
    torch.cuda.empty_cache()
    scan_us = scan_ms / max(n_scan, 1) * 1e3
    return {
        "metric": "chunk-vectors scanned/sec (whole job)", "value": total * args.c4_steps / elapsed, "unit": "rows/s",
        "ms_per_query": elapsed / args.c4_steps * 1e3, "scaling": "strong", "n_gpus": world, "steps": args.c4_steps,
        "config": {"workload": f"c4: 1 query x {total // 1_000_000}M chunks (D=256, f32) row-sharded over {world} GPU(s), "
                               f"{per / 1e6:g}M/GPU, per-shard top-{k} + RCCL all-gather + merge" if world > 1 or group is not None
                               else f"c4: 1 query x {total // 1_000_000}M chunks (D=256, f32) on ONE GPU ({per / 1e6:g}M/GPU, no exchange), top-{k}",
                   "rows_total": total, "rows_per_gpu": per},
        "roofline": {"kernel": "scan_topk_kernel (K2), rank 0's shard", "bound": "hbm",
                     "achieved": (my_rows * ROW_BYTES / (scan_us * 1e-6) / 1e9) if n_scan else None, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": (my_rows * ROW_BYTES / (scan_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if n_scan else None,
                     "avg_kernel_us": scan_us, "launches": n_scan, "algorithmic_bytes_per_launch": my_rows * ROW_BYTES,
                     "traffic": measured_traffic("c4", my_rows)[0], "traffic_source": measured_traffic("c4", my_rows)[1]},
        "operand_image": image_leg,
        "checks": {"torch_fp64_topk_distances_match": ok, "rows_in_range": rows_ok, "rows_match_fp64_topk": rows_match,
                   "selects_without_exactness_certificate": uncertain},
    }


def bench_c4_one_process(smt, args, devs, group, local_ctxs, k, queries_on, host, n_gpus):
    """bench_c4 for the one-process group: the 100M rows cut into len(devs) contiguous shards, shard i generated on local device i,
    one ShardedCorpus, ONE answer per query delivered by local device 0 into pinned memory."""
    import ctypes as C

    from semtools_amd import _lib as L
    total = args.c4_rows
    n = len(devs)
    per = -(-total // n)
    sizes = [max(0, min(per, total - i * per)) for i in range(n)]
    xs = []
    for i, d in enumerate(devs):
        x = torch.empty((sizes[i], 256), device=d, dtype=torch.float32)
        g = torch.Generator(device=d)
        g.manual_seed(3 + i)
        for b in range(0, sizes[i], 2_000_000):
            e = min(sizes[i], b + 2_000_000)
            c = torch.randn(e - b, 256, device=d, generator=g)
            c /= c.norm(dim=1, keepdim=True)
            x[b:e] = c
            del c
        xs.append(x)
    for d in set(devs):
        torch.cuda.synchronize(d)
    corpus = smt.ShardedCorpus(group, device_ptrs=[x.data_ptr() if x.numel() else 0 for x in xs], shard_rows=sizes)
    ring, nq16 = host.shape[0], queries_on[0].shape[0]
    q_arrays = [(C.c_void_p * n)(*[C.c_void_p(queries_on[i][j].data_ptr()) for i in range(n)]) for j in range(nq16)]
    o_arrays = [(C.c_void_p * n)(*([C.c_void_p(host[r].data_ptr())] + [C.c_void_p(None)] * (n - 1))) for r in range(ring)]
    fn = L.lib().smt_sharded_search_topk_device

    def one(i):
        L.check(fn(corpus._h, q_arrays[i % nq16], 1, k, o_arrays[i % ring]))

    ctx = local_ctxs[0]
    for c in local_ctxs:
        c.set_tuning("async_select", 1)
    ctx.set_tuning("prof_select", 0)
    ctx.set_tuning("prof_every", 1)
    for i in range(3):
        one(i)
    group.synchronize()
    ctx.prof_enable(True)
    ctx.prof_reset()
    for c in local_ctxs:
        c.uncertain_count()
    group.synchronize()
    t0 = time.perf_counter()
    for i in range(args.c4_steps):
        one(i)
    group.synchronize()
    elapsed = time.perf_counter() - t0
    n_scan, scan_ms = ctx.prof_read("scan")
    ctx.prof_enable(False)
    for c in local_ctxs:
        c.set_tuning("async_select", 0)
    ctx.set_tuning("prof_select", 1)
    uncertain = sum(c.uncertain_count() for c in local_ctxs)
    last = args.c4_steps - 1
    got_dist = host[last % ring, 1].view(torch.float64).numpy().copy()
    got_rows = host[last % ring, 0].numpy().copy()
    cv, cr = [], []
    for i in range(n):
        if not sizes[i]:
            continue
        lv, li = torch_topk_fp64(xs[i], queries_on[i][last % nq16], k)
        cv.append(lv.cpu())
        cr.append(li.cpu() + i * per)
    allv, allr = torch.cat(cv).numpy(), torch.cat(cr).numpy()
    order = np.lexsort((allr, allv))[:k]
    ok = bool(np.allclose(got_dist, allv[order], rtol=0, atol=1e-6))
    rows_match = bool(got_rows.tolist() == allr[order].tolist())
    rows_ok = bool(((got_rows >= 0) & (got_rows < total)).all())
    corpus.close()
    del xs
    torch.cuda.empty_cache()
    scan_us = scan_ms / max(n_scan, 1) * 1e3
    frac = (sizes[0] * ROW_BYTES / (scan_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if n_scan else None
    return {
        "metric": "chunk-vectors scanned/sec (whole job)", "value": total * args.c4_steps / elapsed, "unit": "rows/s",
        "ms_per_query": elapsed / args.c4_steps * 1e3, "scaling": "strong", "n_gpus": n_gpus, "steps": args.c4_steps,
        "config": {"workload": f"c4: 1 query x {total // 1_000_000}M chunks (D=256, f32) row-sharded over {n} shard(s) of ONE process' group, "
                               f"{per / 1e6:g}M per shard, per-shard top-{k} + '{group.transport}' exchange + merge on device 0",
                   "rows_total": total, "rows_per_gpu": per},
        "roofline": {"kernel": "scan_topk_kernel (K2), shard 0", "bound": "hbm",
                     "achieved": (frac * HBM_PEAK_GBPS) if frac else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": frac,
                     "avg_kernel_us": scan_us, "launches": n_scan, "algorithmic_bytes_per_launch": sizes[0] * ROW_BYTES,
                     "traffic": None, "traffic_source": None},
        "operand_image": None,
        "checks": {"torch_fp64_topk_distances_match": ok, "rows_in_range": rows_ok, "rows_match_fp64_topk": rows_match,
                   "selects_without_exactness_certificate": uncertain},
    }


def bench_c5_sharded(smt, args, devs, rank, n_shards, group, k, nq=1000, nlist=4096, nprobe=8, rerank=128):
    """BASELINE config c5 on the job's GPUs (N > 1): --c5-rows-total rows (default 100 M) of the clustered corpus row-sharded over the
    group -- shard s generated on its device from seed 12 + s --, ONE IVF index with shared centroids (the coarse k-means is
    data-parallel: fixed-point centroid sums all-reduced per iteration; codes fitted per shard), searched through the group's exchange;
    recall@k against the exact sharded search of the same rows.  SPMD: every process makes the same calls; rank 0 reports."""
    from tests import synth

    total = args.c5_rows_total
    per = -(-total // n_shards)
    first_shard = 0 if len(devs) > 1 else rank                 # one process: shards 0 .. n-1; one rank per process: shard = rank
    xs, sizes = [], []
    for i, d in enumerate(devs):
        sidx = first_shard + i
        n_rows = max(0, min(per, total - sidx * per))
        gen = synth.clustered_model_torch(20000, 8, 11, d)     # (the same generative model on every device)
        xs.append(synth.clustered_sample_torch(gen, n_rows, 12 + sidx))
        sizes.append(n_rows)
        if i == 0:
            q = synth.clustered_sample_torch(gen, nq, 13).cpu().numpy()
        del gen
    for d in set(devs):
        torch.cuda.synchronize(d)
    sc = smt.ShardedCorpus(group, device_ptrs=[x.data_ptr() for x in xs], shard_rows=sizes)
    t0 = time.perf_counter()
    exact = sc.search(q, top_k=k)
    exact_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    six = smt.ShardedIvfPq(sc, nlist=nlist, train_iters=10, local_pca=True, shared_centroids=True)
    build_s = time.perf_counter() - t0
    six.search(q, top_k=k, nprobe=nprobe, rerank=rerank)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        got = six.search(q, top_k=k, nprobe=nprobe, rerank=rerank)
    dt = (time.perf_counter() - t0) / reps
    hit = sum(len(set(r.tolist()) & set(e.tolist())) for (r, _), (e, _) in zip(got, exact))
    rows_global = all(int(r.max()) < total for r, _ in got if len(r))
    ascending = all(bool((np.diff(d) >= 0).all()) for _, d in got)
    info = six.info()
    six.close()
    sc.close()
    del xs
    torch.cuda.empty_cache()
    return {"config": {"workload": f"c5 sharded: IVF index nlist={nlist} with shared centroids, 32 B codes per row, over {total} chunks row-sharded over "
                                   f"{n_shards} ranks ({per} per rank), {nq} independent queries through the host form, nprobe={nprobe}, {rerank} re-scored, top-{k}"},
            "rows_total": total, "rows_per_rank": per, "n_shards": n_shards, "build_s": build_s, "index_bytes_local": info["index_bytes"],
            "recall_at_k_vs_exact_sharded_search": hit / (nq * k), "queries_per_s": nq / dt, "ms_per_batch": dt * 1e3, "exact_batch_search_s": exact_s,
            "checks": {"rows_are_global": bool(rows_global), "distances_ascending": bool(ascending)}}


def bench_group_issue(smt, device, n_shards=8, rows=1_000_000, k=10, n=40):
    """What ONE host thread pays to issue a 1-query search over n_shards row shards (the caller of the library is one
    synchronous thread: src/bin/semtools.rs:134-135) and receive ONE answer (on local device 0, as smt_sharded_search does).  On
    one GPU the shards are logical ranks of this device (their GPU work serialises): the figure of interest is the HOST time inside
    smt_sharded_search_topk_device, not the end-to-end time.  Default transport of a one-process group: PEER -- the merge kernel
    reads the ranks' k-lists in place, ordered by one event per rank (a group of real GPUs does exactly the same over xGMI).
    Beside it: the copy transport (gather into rank 0's block), every rank wanting the answer, and the same thread issuing the
    shards' scan + select launches itself, one after the other."""
    import ctypes as C

    from semtools_amd import _lib as L
    g = torch.Generator(device=device)
    g.manual_seed(3)
    shards = []
    for _ in range(n_shards):
        x = torch.randn(rows, 256, device=device, generator=g)
        x /= x.norm(dim=1, keepdim=True)
        shards.append(x)
    q = torch.randn(16, 256, device=device, generator=g)
    q /= q.norm(dim=1, keepdim=True)
    torch.cuda.synchronize(device)
    grp = smt.Group.logical(device.index or 0, n_shards)
    sc = smt.ShardedCorpus(grp, device_ptrs=[sh.data_ptr() for sh in shards], shard_rows=[rows] * n_shards)
    outs = [torch.empty((2, k), dtype=torch.int64, device=device) for _ in range(n_shards)]
    qp = [(C.c_void_p * n_shards)(*[C.c_void_p(q[j].data_ptr())] * n_shards) for j in range(16)]
    op_all = (C.c_void_p * n_shards)(*[C.c_void_p(o.data_ptr()) for o in outs])
    op_one = (C.c_void_p * n_shards)(*([C.c_void_p(outs[0].data_ptr())] + [C.c_void_p(None)] * (n_shards - 1)))
    fn = L.lib().smt_sharded_search_topk_device
    allx = torch.cat(shards)
    d = 1.0 - (allx.double() @ q[(n - 1) % 16].double())
    truth = torch.topk(d, k, largest=False)[1].cpu().tolist()
    del allx, d

    def run(op, rounds=5):
        # the MEDIAN of `rounds` rounds of n searches: 4 ms of one host thread's time per round is at the mercy of whatever else the
        # host does (the same box read 70 and 140 us in two bench runs minutes apart when this was a single round)
        issued_us, total_us = [], []
        for _ in range(rounds):
            for j in range(8):
                L.check(fn(sc._h, qp[j % 16], 1, k, op))
            grp.synchronize()
            t0 = time.perf_counter()
            for j in range(n):
                fn(sc._h, qp[j % 16], 1, k, op)
            issued = time.perf_counter() - t0
            grp.synchronize()
            total = time.perf_counter() - t0
            issued_us.append(issued / n * 1e6)
            total_us.append(total / n * 1e6)
        return sorted(issued_us)[rounds // 2], sorted(total_us)[rounds // 2], bool(outs[0].cpu().numpy()[0].tolist() == truth)

    default_transport = grp.transport
    peer_us, peer_e2e, ok_peer = run(op_one)
    peer_all_us, _, ok_peer_all = run(op_all)
    grp.set_transport("copy")
    copy_us, _, ok_copy = run(op_one)
    copy_all_us, _, ok_copy_all = run(op_all)
    grp.set_transport(default_transport)
    views = [sc.shard(i)[0] for i in range(n_shards)]
    o_r = [torch.empty(k, dtype=torch.int64, device=device) for _ in range(n_shards)]
    o_d = [torch.empty(k, dtype=torch.float64, device=device) for _ in range(n_shards)]
    fn1 = L.lib().smt_search_topk_device
    args1 = [(views[i]._h, C.c_void_p(q[0].data_ptr()), 1, k, 0, C.c_void_p(o_r[i].data_ptr()), C.c_void_p(o_d[i].data_ptr())) for i in range(n_shards)]
    for a in args1:
        L.check(fn1(*a))
    grp.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for a in args1:
            fn1(*a)
    serial = (time.perf_counter() - t0) / n
    grp.synchronize()
    sc.close()
    grp.close()
    del shards
    torch.cuda.empty_cache()
    return {"metric": "host microseconds to issue one sharded search (one caller thread, one answer)", "shards": n_shards, "rows_per_shard": rows,
            "transport": f"{default_transport} (default of a one-process group): logical ranks on one device, merge kernel reads the ranks' k-lists "
                         "in place, one event per rank; the RCCL collective itself cannot run on one GPU",
            "host_issue_us_per_search": peer_us, "end_to_end_us_per_search": peer_e2e,
            "every_rank_wants_the_answer_us": peer_all_us,
            "copy_transport_us": copy_us, "copy_transport_every_rank_us": copy_all_us,
            "one_thread_issues_every_shard_us": serial * 1e6, "one_shard_scan_us": 150.0,
            "checks": {"last_answer_matches_fp64_topk": ok_peer and ok_peer_all, "copy_transport_answer_matches": ok_copy and ok_copy_all}}


def bench_small_calls(smt, ctx, device, shard, k):
    """Latency of the HOST-form call (smt_search: host query in, host answer out, synchronous) where the reference lives: c1 -- one
    query over 1000 lines, top-3 (src/cmds/search.rs:245-257; the agent tool repeats it, src/ask/tools.rs:229-258) -- beside the
    call's fixed cost (one query over ONE row) and the 1 M-row call of c2.  Small answers are DELIVERED by the select kernel into
    pinned host memory (tuning key direct_delivery); the same calls with the key off show what the D2H copy + hipStreamSynchronize
    cost.  Median of 7 rounds of 200 calls, microseconds per call; every answer checked against the fp64 reference of torch."""
    import ctypes as C

    from semtools_amd import _lib as L

    g = torch.Generator(device=device)
    g.manual_seed(77)
    small = torch.randn(65536, 256, device=device, generator=g)
    small /= small.norm(dim=1, keepdim=True)
    q = torch.randn(4, 256, device=device, generator=g)
    q /= q.norm(dim=1, keepdim=True)
    qh = np.ascontiguousarray(q.cpu().numpy())
    torch.cuda.synchronize(device)
    o_rows = np.empty((4, 16), dtype=np.uint64)
    o_dist = np.empty((4, 16), dtype=np.float64)
    o_cnt = np.zeros(4, dtype=np.uint64)

    def call(corpus, nq, kk):
        L.check(L.lib().smt_search(corpus._h, L.np_ptr(qh), nq, kk, float("nan"), L.MODE_DOCUMENTS, None, 0, 0,
                                   L.np_ptr(o_rows), L.np_ptr(o_dist), L.np_ptr(o_cnt), 16))

    def timed(corpus, nq, kk, reps=200, rounds=7):
        for _ in range(20):
            call(corpus, nq, kk)
        v = []
        for _ in range(rounds):
            t0 = time.perf_counter()
            for _ in range(reps):
                call(corpus, nq, kk)
            v.append((time.perf_counter() - t0) / reps * 1e6)
        return float(np.median(v))

    def check(corpus_rows, n_rows, nq, kk):
        sim = (q[:nq].double() @ corpus_rows[:n_rows].double().T)
        nrm = corpus_rows[:n_rows].double().norm(dim=1)
        d = (1.0 - sim / (q[:nq].double().norm(dim=1)[:, None] * nrm[None, :])).clamp_min(0)
        want = torch.topk(d, min(kk, n_rows), dim=1, largest=False, sorted=True)
        ok = True
        for i in range(nq):
            n = int(o_cnt[i])
            ok = ok and n == min(kk, n_rows) and o_rows[i, :n].astype(np.int64).tolist() == want.indices[i].cpu().tolist()
            ok = ok and bool(np.abs(o_dist[i, :n] - want.values[i].cpu().numpy()).max() < 1e-9)
        return ok

    one = smt.Corpus(ctx, device_ptr=small.data_ptr(), rows=1)
    c1 = smt.Corpus(ctx, device_ptr=small.data_ptr(), rows=1000)
    c64k = smt.Corpus(ctx, device_ptr=small.data_ptr(), rows=65536)
    c2 = smt.Corpus(ctx, device_ptr=shard.data_ptr(), rows=shard.shape[0])
    out = {"config": {"workload": "smt_search, host buffers in and out, one synchronous call at a time: us per call (median of 7 x 200 calls)"}}
    checks = {}
    for fused in (1, 0):
        ctx.set_tuning("direct_delivery", fused)
        tag = "delivered" if fused else "copy_and_synchronize"
        leg = {}
        leg["one_row_1q_us"] = timed(one, 1, 3)
        leg["c1_1000_rows_1q_top3_us"] = timed(c1, 1, 3)
        call(c1, 1, 3)
        checks[f"c1_answer_{tag}"] = check(small, 1000, 1, 3)
        leg["c1_1000_rows_3q_top3_us"] = timed(c1, 3, 3)
        call(c1, 3, 3)
        checks[f"c1_3q_answer_{tag}"] = check(small, 1000, 3, 3)
        leg["rows_65536_1q_top10_us"] = timed(c64k, 1, 10)
        call(c64k, 1, 10)
        checks[f"64k_answer_{tag}"] = check(small, 65536, 1, 10)
        out[tag] = leg
        leg["c2_1M_rows_1q_us"] = timed(c2, 1, k, reps=100)
    ctx.set_tuning("direct_delivery", 1)
    out["deliveries"] = ctx.deliveries()
    out["checks"] = checks
    for c in (one, c1, c64k, c2):
        c.close()
    return out


def bench_workspace(smt, ctx, device, rows, k, nq_batch=256, n_docs=10_000, max_d=0.9):
    """Workspace mode (Store::search_line_embeddings, src/workspace/store.rs:481-546: path-subset filter :507-515, score threshold
    1 - max_distance :502-503, `limit` then truncate(top_k) :543) through smt_search: `rows` line embeddings in n_docs "documents",
    the subset = every second document (n_docs / 2 row ranges).  One query (K2 over the chunk table; over the operand image once the
    corpus has one) and a batch (gemm_rowreg_kernel over the tile table), each against the SAME call without the filter.  Whole host
    calls: queries and ranges go in as host arrays, hits come back as host arrays -- what the store pays.  Algorithmic bytes: the
    SCANNED rows x 1 KiB (f32 rows) resp. 512 B (image)."""
    from oracle import oracle as orc

    g = torch.Generator(device=device)
    g.manual_seed(3)
    x = torch.empty((rows, 256), device=device, dtype=torch.float32)
    for b in range(0, rows, 2_000_000):
        e = min(rows, b + 2_000_000)
        c = torch.randn(e - b, 256, device=device, generator=g)
        c /= c.norm(dim=1, keepdim=True)
        x[b:e] = c
    del c
    g.manual_seed(6)
    qd = torch.randn(nq_batch, 256, device=device, generator=g)
    qd /= qd.norm(dim=1, keepdim=True)
    q = qd.cpu().numpy()
    per = rows // n_docs
    ranges = [(d * per, (d + 1) * per) for d in range(0, n_docs, 2)]
    packed = smt.PackedRanges(ranges)
    scanned = sum(e - b for b, e in ranges)
    corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
    kw = dict(top_k=k, max_distance=max_d, mode=smt.MODE_WORKSPACE)

    # smt_search called the way a C / Rust host calls it: host arrays in, preallocated host arrays out (the Python wrapper's
    # per-query list building -- ~0.1 ms at 256 queries -- is not the library's work)
    import ctypes as C

    from semtools_amd import _lib as L
    o_rows = np.empty((nq_batch, k), dtype=np.uint64)
    o_dist = np.empty((nq_batch, k), dtype=np.float64)
    o_cnt = np.zeros(nq_batch, dtype=np.uint64)

    def call(queries, filtered):
        n = queries.shape[0]
        L.check(L.lib().smt_search(corpus._h, L.np_ptr(queries), n, k, float(max_d), smt.MODE_WORKSPACE,
                                   C.cast(packed.arr, C.c_void_p) if filtered else None, packed.n if filtered else 0, 0,
                                   L.np_ptr(o_rows), L.np_ptr(o_dist), L.np_ptr(o_cnt), k))
        return [(o_rows[i, :int(o_cnt[i])].copy(), o_dist[i, :int(o_cnt[i])].copy()) for i in (0, n - 1)]

    flagged = []

    def timed(queries, filtered, reps, prof):
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        ctx.uncertain_count()
        call(queries, filtered)
        call(queries, filtered)
        ctx.synchronize()

        def loop():
            for _ in range(reps):
                L.lib().smt_search(corpus._h, L.np_ptr(queries), queries.shape[0], k, float(max_d), smt.MODE_WORKSPACE,
                                   C.cast(packed.arr, C.c_void_p) if filtered else None, packed.n if filtered else 0, 0,
                                   L.np_ptr(o_rows), L.np_ptr(o_dist), L.np_ptr(o_cnt), k)

        # whole calls timed WITHOUT the profiling events (each bracketed launch costs ~5 us of stream time: 40 us of a 0.5 ms one-query
        # call over the image, where eight kernels are bracketed); the kernel time comes from a second, profiled loop
        t0 = time.perf_counter()
        loop()
        dt = (time.perf_counter() - t0) / reps
        ctx.set_tuning("prof_every", 1)
        ctx.prof_enable(True)
        ctx.prof_reset()
        loop()
        n, ms = ctx.prof_read(prof)
        ctx.prof_enable(False)
        flagged.append(int(ctx.uncertain_count()))   # selects whose certificate failed (re-answered exhaustively by the host call)
        got = call(queries, filtered)
        return dt, (ms / reps * 1e-3) if n else None, n // reps, got

    def one_leg(image):
        # one query
        t1, ker1, n1, got1 = timed(q[:1], True, 20, "gemm" if image else "scan")
        t1u, _, _, _ = timed(q[:1], False, 20, "gemm" if image else "scan")
        bytes_row = 512 if image else ROW_BYTES
        one = {"ms_per_call": t1 * 1e3, "rows_per_s": scanned / t1, "unfiltered_ms_per_call": t1u * 1e3,
               "kernel_ms_per_call": ker1 * 1e3 if ker1 else None, "kernel_launches_per_call": n1}
        if image:
            one["frac_of_hbm_at_512B_per_row"] = scanned * 512 / t1 / (HBM_PEAK_GBPS * 1e9)
            one["note"] = "whole call over the fp16 operand image (gemm_rowreg_kernel<.., true> over the tile table: levels, selects, delivery)"
        else:
            ach = scanned * ROW_BYTES / ker1 / 1e9 if ker1 else None
            one["roofline"] = {"kernel": "scan_topk_kernel<1, 4, nt, FILTERED> (K2 over the chunk table)", "bound": "hbm", "achieved": ach,
                               "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS if ach else None,
                               "traffic": measured_traffic("workspace_one_query", scanned)[0], "traffic_source": measured_traffic("workspace_one_query", scanned)[1],
                               "algorithmic_bytes_per_launch": scanned * ROW_BYTES, "launches": n1 * 20}
        # a batch
        tb, kerb, nb, gotb = timed(q, True, 5, "gemm")
        tbu, kerbu, _, _ = timed(q, False, 5, "gemm")
        batch = {"queries": nq_batch, "ms_per_call": tb * 1e3, "queries_per_s": nq_batch / tb, "gemm_ms_per_call": kerb * 1e3 if kerb else None,
                 "gemm_launches_per_call": nb, "unfiltered_ms_per_call": tbu * 1e3, "unfiltered_gemm_ms_per_call": kerbu * 1e3 if kerbu else None,
                 "cost_per_scanned_row_vs_unfiltered": (tb / scanned) / (tbu / rows),
                 "gemm_cost_per_scanned_row_vs_unfiltered": ((kerb / scanned) / (kerbu / rows)) if kerb and kerbu else None,
                 "scanned_bytes_per_s": scanned * bytes_row / tb}
        return one, batch, got1, gotb

    ctx.uncertain_count()
    one, batch, got1, gotb = one_leg(False)
    t0 = time.perf_counter()
    corpus.prepack()
    ctx.synchronize()
    prepack_s = time.perf_counter() - t0
    one_i, batch_i, got1_i, gotb_i = one_leg(True)
    same_1 = got1[0][0].tolist() == got1_i[0][0].tolist() and bool(np.array_equal(got1[0][1], got1_i[0][1]))
    # every query of the batch, image against f32 rows (through the wrapper, outside the timed loops)
    full_i = corpus.search(q, ranges=packed, **kw)
    corpus.prepack(False)
    full_f = corpus.search(q, ranges=packed, **kw)
    same_b = sum(int(a[0].tolist() == b[0].tolist() and np.array_equal(a[1], b[1])) for a, b in zip(full_f, full_i))
    # ---- the last answers against (1) an independent fp64 top-k over every eligible row on the device and (2) the CPU oracle's
    # store search (orc_search_line_embeddings: path filter, f32 score threshold, order, truncate) over the 4096 fp64-nearest
    # eligible rows of query 0 -- the whole 5 M-row subset would be a 5 GB copy and seconds of scalar code
    starts = torch.arange(0, n_docs, 2, device=device, dtype=torch.int64) * per
    elig = (starts[:, None] + torch.arange(per, device=device, dtype=torch.int64)[None, :]).reshape(-1)
    thr = float(np.float32(1.0) - np.float32(max_d))

    def fp64_answer(qv, n_keep):
        best_v = best_i = None
        for b in range(0, elig.numel(), 1_000_000):
            idx = elig[b:b + 1_000_000]
            dd = 1.0 - (x[idx].double() @ qv.double())
            v, i = torch.topk(dd, min(n_keep, dd.numel()), largest=False)
            i = idx[i]
            if best_v is not None:
                v, i = torch.cat([best_v, v]), torch.cat([best_i, i])
                v, sel = torch.topk(v, min(n_keep, v.numel()), largest=False)
                i = i[sel]
            best_v, best_i = v, i
        order = torch.argsort(best_v, stable=True)
        return best_v[order], best_i[order]

    fv, fi = fp64_answer(qd[0], 4096)
    keep = (1.0 - fv[:k]) > thr
    rows_ok_1 = got1[0][0].tolist() == fi[:k][keep].cpu().tolist() and bool(np.allclose(got1[0][1], fv[:k][keep].cpu().numpy(), rtol=0, atol=1e-6))
    fvb, fib = fp64_answer(qd[nq_batch - 1], k)
    keepb = (1.0 - fvb) > thr
    rows_ok_b = gotb[-1][0].tolist() == fib[keepb].cpu().tolist()
    cand = torch.sort(fi)[0]
    cand_np = cand.cpu().numpy()
    res = orc.search_line_embeddings(x[cand].cpu().numpy(), (cand_np // per).astype(np.uint32), (cand_np % per).astype(np.int32), q[0],
                                     np.arange(0, n_docs, 2, dtype=np.uint32), k, max_d)
    orc_rows = [r["path_id"] * per + r["line_number"] for r in res]
    orc_d = np.array([r["distance"] for r in res], dtype=np.float64)
    oracle_rows = orc_rows == got1[0][0].tolist()
    oracle_diff = float(np.abs(orc_d - got1[0][1]).max()) if oracle_rows and len(orc_rows) else (0.0 if oracle_rows else float("inf"))
    uncertain = ctx.uncertain_count()
    corpus.close()
    del x, elig
    torch.cuda.empty_cache()
    return {"metric": "workspace-mode searches over a document subset (A10)", "rows": rows, "documents": n_docs, "ranges": len(ranges),
            "rows_scanned": scanned, "top_k": k, "max_distance": max_d,
            "config": {"workload": f"workspace: {rows} line embeddings in {n_docs} documents, subset = every second document "
                                   f"({len(ranges)} row ranges, {scanned} rows), mode 1, max_distance {max_d}, top-{k}; 1 query and {nq_batch} queries"},
            "one_query": one, "batch": batch, "one_query_image": one_i, "batch_image": batch_i, "image_build_ms": prepack_s * 1e3,
            "checks": {"one_query_rows_match_fp64_topk": rows_ok_1, "batch_last_query_rows_match_fp64_topk": rows_ok_b,
                       "oracle_store_search_rows_match": oracle_rows, "oracle_store_search_dist_max_abs_diff": oracle_diff,
                       "image_and_f32_rows_agree_one_query": same_1, "image_and_f32_rows_agree_batch": f"{same_b}/{nq_batch}",
                       "selects_without_exactness_certificate_in_timed_calls": int(sum(flagged))}}


def bench_ingest(smt, ctx, n_lines, vocab=50_000):
    """The host step in FRONT of the path (src/search/mod.rs:49-75 create_document_from_content -> :69 encode_with_args): one file of
    n_lines lines of pseudo-prose goes through the C++ host layer -- split into lines, tokenise on the host cores, upload ids, K1 --
    pipelined (tokenise || H2D || K1), then one search.  lines/s of the whole call, with the whitespace-hash tokenizer (the floor of the
    host layer itself) and with a WordPiece tokenizer.json read by the native tokenizer (what a model2vec English model runs)."""
    import ctypes as C
    import tempfile

    from semtools_amd import _lib as L
    from semtools_amd import host
    from tests import synth

    table = synth.table(vocab, seed=2)
    lines = synth.pseudo_prose(20_000, vocab_size=vocab, seed=1)
    content = "\n".join(lines[i % len(lines)] for i in range(n_lines)) + "\n"
    n_tok = sum(len(ln.split()) for ln in lines) / len(lines) * n_lines
    content_b, query_b = content.encode(), lines[17].encode()     # (Python's str -> bytes copy is not the host layer's work)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

    def run(model):
        def phases_now():
            try:   # PhaseTimer of the host layer: cumulative over the process
                return json.loads(host._take_text(L.lib().smt_host_timing_json()) or "{}")
            except Exception:
                return None
        before = phases_now() or {}
        best, first = None, ""
        for _ in range(3):
            out = C.c_void_p()
            t0 = time.perf_counter()
            L.check(L.lib().smt_host_search_content(model._h, query_b, b"<stdin>", content_b, 0, 3, float("nan"), 0, 0, 0, C.byref(out)))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            first = host._take_text(out).split("\n")[0]
        after = phases_now()
        phases = None if after is None else {k: round(v - before.get(k, 0.0), 3) for k, v in after.items()
                                             if k != "between_calls" and v - before.get(k, 0.0) > 0}
        try:
            first_d = float(first[first.rindex("(") + 1:first.rindex(")")])
        except Exception:
            first_d = float("nan")
        return {"seconds": best, "lines_per_s": n_lines / best, "text_MB_per_s": len(content) / best / 1e6,
                "host_phases_ms_over_3_calls": phases, "first_hit": first[:80], "first_hit_is_the_query_line": bool(first_d < 1e-6)}

    model = host.StaticModel(ctx, table=table, tokenizer="hash")
    r = run(model)
    model.close()
    res = {"metric": "lines ingested/sec (split + tokenise + upload + K1 + one search, one MI355X)", "lines": n_lines, "text_bytes": len(content),
           "tokens": int(n_tok), "seconds": r["seconds"], "lines_per_s": r["lines_per_s"], "text_MB_per_s": r["text_MB_per_s"],
           "tokens_per_s": n_tok / r["seconds"], "cores": cores, "host_cpu": _cpu_model(), "tokenizer": "whitespace-hash (host threads)",
           "host_phases_ms_over_3_calls": r["host_phases_ms_over_3_calls"], "first_hit": r["first_hit"],
           "checks": {"first_hit_is_the_query_line": r["first_hit_is_the_query_line"]}}
    # ---- the same file through a real tokenizer.json: WordPiece + BertNormalizer + BertPreTokenizer, trained here on the same prose
    # (no network: the `tokenizers` wheel only BUILDS the file; the library reads and runs it natively, hf_tokenizer.cpp)
    try:
        from safetensors.numpy import save_file
        from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, trainers
        tok = Tokenizer(models.WordPiece(unk_token="[UNK]"))
        tok.normalizer = normalizers.BertNormalizer(lowercase=True)
        tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
        tok.train_from_iterator(lines, trainers.WordPieceTrainer(vocab_size=8000, special_tokens=["[PAD]", "[UNK]"], show_progress=False))
        with tempfile.TemporaryDirectory() as d:
            tok.save(os.path.join(d, "tokenizer.json"))
            save_file({"embeddings": synth.table(tok.get_vocab_size(), seed=5)}, os.path.join(d, "model.safetensors"))
            with open(os.path.join(d, "config.json"), "w") as fh:
                json.dump({"normalize": True}, fh)
            model = host.StaticModel(ctx, model_dir=d)
            w = run(model)
            model.close()
        sample = lines[:2000]
        w_tok = sum(len(e.ids) for e in tok.encode_batch(sample, add_special_tokens=False)) / len(sample) * n_lines
        res["wordpiece_tokenizer_json"] = {"tokenizer": "WordPiece 8 k pieces, BertNormalizer, BertPreTokenizer (native reader, host threads)",
                                           "tokens": int(w_tok), "seconds": w["seconds"], "lines_per_s": w["lines_per_s"],
                                           "text_MB_per_s": w["text_MB_per_s"], "tokens_per_s": w_tok / w["seconds"],
                                           "host_phases_ms_over_3_calls": w["host_phases_ms_over_3_calls"], "first_hit": w["first_hit"]}
        res["checks"]["wordpiece_first_hit_is_the_query_line"] = w["first_hit_is_the_query_line"]
    except Exception as exc:   # (a secondary figure: the wheel missing or behaving differently must not take the hash-tokenizer figure down)
        res["wordpiece_tokenizer_json"] = {"skipped": repr(exc)}
    return res


def bench_embed(smt, ctx, device, n_lines, vocab=500_000, reps=5):
    """K1 (embed: token-id gather from the embedding table + mean-pool + L2-normalise, src/search/mod.rs:69 ->
    model2vec-rs encode_with_args' pool step): n_lines ragged lines (0..32 tokens) over a V x 256 f32 table the size of
    potion-multilingual-128M's (500 k rows, 512 MB), ids and output resident.  Algorithmic bytes (SURVEY 8a A3 / DESIGN 4.4):
    1 KiB gathered per token + 1 KiB written per line.  Two id distributions: Zipf (natural text: hot rows stay in L2 / MALL,
    the rate can exceed HBM's) and UNIFORM over a 4 M-row (4 GB) table (no cache help: the HBM-bound figure, the one the
    roofline fraction is quoted on).  A sample of lines is compared bit for bit with the CPU oracle."""
    from tests import synth

    legs = {}
    for name, V, uniform in (("zipf_ids_500k_table", vocab, False), ("uniform_ids_4M_table", 4_000_000, True)):
        g = torch.Generator(device=device)
        g.manual_seed(2)
        table = torch.randn(V, 256, device=device, generator=g) * 0.1
        ids, offsets = synth.token_lines(n_lines, V=V if not uniform else vocab, seed=1, min_tok=0, max_tok=32)
        if uniform:
            ids = np.random.default_rng(7).integers(0, V, size=ids.size).astype(np.uint32)
        d_ids = torch.from_numpy(ids.astype(np.int32)).to(device)
        d_off = torch.from_numpy(offsets.astype(np.int64)).to(device)
        out = torch.empty(n_lines, 256, device=device)
        model = smt.Model(ctx, device_ptr=table.data_ptr(), V=V, normalize=True)
        ctx.prof_enable(True)
        model.embed_device(d_ids.data_ptr(), d_off.data_ptr(), n_lines, 2048, out.data_ptr())
        torch.cuda.synchronize(device)
        ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(reps):
            model.embed_device(d_ids.data_ptr(), d_off.data_ptr(), n_lines, 2048, out.data_ptr())
        torch.cuda.synchronize(device)
        wall = (time.perf_counter() - t0) / reps
        n, ms = ctx.prof_read("embed")
        ctx.prof_enable(False)
        ker = ms / max(n, 1) * 1e-3
        T = int(ids.size)
        alg = (T + n_lines) * 1024
        # bit-exact check of a sample of lines against the oracle (the table rows they touch are copied back)
        from oracle import oracle as orc
        sample = np.linspace(0, n_lines - 1, 257).astype(np.int64)
        s_ids = np.concatenate([ids[int(offsets[i]):int(offsets[i + 1])] for i in sample]) if len(sample) else np.empty(0, np.uint32)
        s_off = np.concatenate([[0], np.cumsum([int(offsets[i + 1] - offsets[i]) for i in sample])]).astype(np.uint64)
        uniq, inv = np.unique(s_ids, return_inverse=True)
        sub = table[torch.from_numpy(uniq.astype(np.int64)).to(device)].cpu().numpy()
        want = orc.embed_lines(sub, inv.astype(np.uint32), s_off, True, 2048)
        got = out[torch.from_numpy(sample).to(device)].cpu().numpy()
        traffic, traffic_source = measured_traffic("embed_" + name, n_lines)
        # Zipf ids: hot table rows are served by L2 / MALL -- the algorithmic bytes never reach HBM, so the fraction of the HBM peak is
        # quoted on the MEASURED traffic (FETCH_SIZE x2, profiles/traffic.json) and the bound is named for what it is
        if uniform:
            roof = {"kernel": "embed_kernel (K1)", "bound": "hbm", "achieved": alg / ker / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": alg / ker / 1e9 / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": alg, "launches": n}
        else:
            roof = {"kernel": "embed_kernel (K1)", "bound": "l2+mall (hot rows) over hbm", "achieved": (traffic / ker / 1e9) if traffic else None,
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": (traffic / ker / 1e9 / HBM_PEAK_GBPS) if traffic else None, "traffic": traffic,
                    "traffic_source": traffic_source, "algorithmic_bytes_per_launch": alg, "algorithmic_rate_GBps": alg / ker / 1e9,
                    "note": "achieved / frac are MEASURED HBM bytes per second; the gathers the caches serve make the algorithmic rate exceed HBM's",
                    "launches": n}
        legs[name] = {
            "lines": n_lines, "tokens": T, "table_rows": V, "kernel_ms": ker * 1e3, "wall_ms": wall * 1e3,
            "lines_per_s": n_lines / ker, "tokens_per_s": T / ker, "roofline": roof,
            "checks": {"sample_lines_bit_exact_vs_oracle": bool(np.array_equal(got, want)), "sample_lines": int(len(sample))},
        }
        model.close()
        del table, d_ids, d_off, out
        torch.cuda.empty_cache()
    return {"metric": "lines embedded/sec (K1, one MI355X, ids resident)", "value": legs["zipf_ids_500k_table"]["lines_per_s"],
            "unit": "lines/s", "config": {"workload": f"K1: {n_lines} ragged lines (0..32 tokens) pooled from a V x 256 f32 table"},
            "roofline": legs["uniform_ids_4M_table"]["roofline"], **legs}


def bench_c3(smt, ctx, device, rows, nq, k, reps=3):
    """Config c3: nq batched queries x rows chunks through the MFMA path (K3: candidates nominated by bf16 x 3 split
    products, answers exact).  queries/sec at a 10M-chunk corpus is the second half of BASELINE.json's metric."""
    g = torch.Generator(device=device)
    g.manual_seed(3)
    x = torch.randn(rows, 256, device=device, generator=g)
    x /= x.norm(dim=1, keepdim=True)
    g.manual_seed(5)
    q = torch.randn(nq, 256, device=device, generator=g)
    q /= q.norm(dim=1, keepdim=True)
    out_rows = torch.empty(nq, k, dtype=torch.int64, device=device)
    out_dist = torch.empty(nq, k, dtype=torch.float64, device=device)
    corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)

    def timed():
        # wall time of whole calls WITHOUT the profiling events (a HIP event pair costs ~5 us of stream time and a batch has eight
        # bracketed launches: until round 5 they sat inside this timed loop), then the same batches again with them for the kernel time
        corpus.search_topk_device(q.data_ptr(), nq, k, 0, out_rows.data_ptr(), out_dist.data_ptr())  # warm-up
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(reps):
            corpus.search_topk_device(q.data_ptr(), nq, k, 0, out_rows.data_ptr(), out_dist.data_ptr())
        torch.cuda.synchronize(device)
        w = (time.perf_counter() - t0) / reps
        ctx.prof_enable(True)
        ctx.prof_reset()
        for _ in range(reps):
            corpus.search_topk_device(q.data_ptr(), nq, k, 0, out_rows.data_ptr(), out_dist.data_ptr())
        torch.cuda.synchronize(device)
        n, ms = ctx.prof_read("gemm")
        ctx.prof_enable(False)
        return w, n, ms / reps * 1e-3

    ctx.uncertain_count()
    # (1) from the f32 rows alone: what a corpus ADOPTED from device memory gets (this one is a torch tensor) until its owner asks
    # for the operand image; (2) with the fp16 operand image (smt_corpus_prepack; a corpus the library owns -- the workspace
    # store's -- builds it at its first batch): half the bytes per row, no row phase.  Same nominations, same answers.
    wall_f32, n_g_f32, gemm_s_f32 = timed()
    rows_f32, dist_f32 = out_rows.clone(), out_dist.clone()
    t0 = time.perf_counter()
    corpus.prepack()
    ctx.synchronize()
    prepack_s = time.perf_counter() - t0
    image_bytes = corpus.image_bytes
    wall, n_g, gemm_s = timed()
    image_same = int(((rows_f32 == out_rows).all(dim=1) & (dist_f32 == out_dist).all(dim=1)).sum().item())

    # one query over the same 10 M rows: the scan kernel over the f32 rows (K2) against the batched kernel over the image, which
    # is what a corpus that has its image does from 1.5 M rows per shard on (topk_dispatch, tuning key image_scan_min_rows)
    def one_query(reps1=20):
        r1 = torch.empty(1, k, dtype=torch.int64, device=device)
        d1 = torch.empty(1, k, dtype=torch.float64, device=device)
        corpus.search_topk_device(q.data_ptr(), 1, k, 0, r1.data_ptr(), d1.data_ptr())
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for _ in range(reps1):
            corpus.search_topk_device(q.data_ptr(), 1, k, 0, r1.data_ptr(), d1.data_ptr())
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t1) / reps1, r1, d1
    ctx.set_tuning("image_scan_min_rows", 0)
    s_f32, r_f32, d_f32 = one_query()
    ctx.set_tuning("image_scan_min_rows", 1_500_000)
    s_img, r_img, d_img = one_query()
    single = {"rows": rows, "f32_scan_ms": s_f32 * 1e3, "f32_scan_rows_per_s": rows / s_f32, "image_ms": s_img * 1e3,
              "image_rows_per_s": rows / s_img, "image_frac_of_hbm_at_512B_per_row": rows * 512 / s_img / 8e12,
              "answers_identical": bool((r_f32 == r_img).all().item()) and bool((d_f32 == d_img).all().item()),
              "note": "whole calls incl. levels, selects and delivery; the c2 / c4 legs (the headline) always scan the f32 rows"}
    flops = 2.0 * nq * rows * 256
    # gemm_topk.hip launch_gemm_topk, the auto rule of gemm_nominate: MFMAs issued per algorithmic multiply-add
    small_shard = rows <= (1 << 25)
    issued_f32 = 1.0 if (nq >= 256 and small_shard and k + 24 <= 64) else 2.0 if (nq >= 128 and small_shard) else 3.0   # from f32 rows
    issued_img = 1.0 if (nq > 96 and k + 24 <= 64) else 2.0 if (k + 16 <= 64) else 3.0                                 # with the image
    names = {1.0: "f16 x 1", 2.0: "f16 x 2", 3.0: "bf16 x 3"}
    ok = True
    for i in range(min(3, nq)):  # independent fp64 check of a few queries
        ref = 1.0 - (x.double() @ q[i].double())
        tv, ti = torch.topk(ref, k, largest=False)
        ok &= bool((out_rows[i] == ti).all().item()) and bool(((out_dist[i] - tv).abs().max() < 1e-6).item())
    # EVERY query of the batch against the single-query scan path (K2, <= 4 queries per pass) on the device:
    # rows identical, distances bit-identical (both paths end in the same exact f64 rescoring)
    k2_rows = torch.empty_like(out_rows)
    k2_dist = torch.empty_like(out_dist)
    ctx.set_tuning("gemm_min_nq", 8)     # (batches of 3 .. 7 queries would take K3 too on a shard this size: keep them on the scan kernel)
    for i in range(0, nq, 4):
        n = min(4, nq - i)
        corpus.search_topk_device(q[i:i + n].data_ptr(), n, k, 0, k2_rows[i:i + n].data_ptr(), k2_dist[i:i + n].data_ptr())
    torch.cuda.synchronize(device)
    ctx.set_tuning("gemm_min_nq", 5)
    same = ((k2_rows == out_rows).all(dim=1) & (k2_dist == out_dist).all(dim=1))
    n_same = int(same.sum().item())
    uncertain = ctx.uncertain_count()
    # The same batch with the score matrix on f32 MFMAs (tuning key gemm_bf16x3 = 0: v_mfma_f32_32x32x2_f32, exact f32
    # products -- "f32 MFMA QxC^T" as BASELINE config c3 words it, SURVEY 8(d) target >= 50 % of 157.3 TF).  Same answers
    # (the scores only nominate); the shipped default above reaches them ~4x sooner on the 16-bit pipe.
    f32_leg = None
    try:
        ctx.set_tuning("gemm_bf16x3", 0)
        f_rows, f_dist = torch.empty_like(out_rows), torch.empty_like(out_dist)
        corpus.search_topk_device(q.data_ptr(), nq, k, 0, f_rows.data_ptr(), f_dist.data_ptr())  # warm-up
        torch.cuda.synchronize(device)
        ctx.prof_enable(True)
        ctx.prof_reset()
        t0 = time.perf_counter()
        corpus.search_topk_device(q.data_ptr(), nq, k, 0, f_rows.data_ptr(), f_dist.data_ptr())
        torch.cuda.synchronize(device)
        f_wall = time.perf_counter() - t0
        n_f, ms_f = ctx.prof_read("gemm")
        ctx.prof_enable(False)
        f_same = int(((f_rows == out_rows).all(dim=1) & (f_dist == out_dist).all(dim=1)).sum().item())
        f32_leg = {"kernel": "gemm_level_kernel (K3, f32 MFMA, tuning key gemm_bf16x3 = 0)", "bound": "mfma",
                   "achieved": flops / (ms_f * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flops / (ms_f * 1e-3) / 157.3e12,
                   "gemm_ms_per_batch": ms_f, "ms_per_batch": f_wall * 1e3, "queries_per_s": nq / f_wall,
                   "agreement_with_default_mode": f"{f_same}/{nq}"}
    except Exception as exc:
        f32_leg = {"error": repr(exc)}
    finally:
        ctx.set_tuning("gemm_bf16x3", 1)
    corpus.close()
    del x
    torch.cuda.empty_cache()
    def k3_roofline(gemm_seconds, launches, leg, bytes_per_row, issued):
        t, src = measured_traffic(leg, rows)
        return {"kernel": f"gemm_rowreg_kernel (K3, {names[issued]})", "bound": "mfma", "achieved": issued * flops / gemm_seconds / 1e12,
                "peak": 2500.0, "unit": "TFLOP/s", "frac": issued * flops / gemm_seconds / 2500e12, "traffic": t, "traffic_source": src,
                "algorithmic_bytes_per_batch": rows * bytes_per_row,
                "algorithmic_bytes_note": "`traffic` is the largest launch of the batch: its last level, 3/4 of the rows",
                "algorithmic_flops_per_batch": flops, "issued_16bit_mfma_flops_per_batch": issued * flops,
                "algorithmic_rate_over_f32_mfma_peak": flops / gemm_seconds / 157.3e12,
                "gemm_ms_per_batch": gemm_seconds * 1e3, "gemm_launches_per_batch": launches,
                # what the 16-bit pipe of this part sustains on DATA (it is power-managed: constants run at 2.4 GHz and 0.98 of
                # nominal, random fp16 operands at ~1.45 GHz): tools/micro/mfma_peak_f16.hip, profiles/r03_mfma_peak_f16.jsonl
                "sustained_peak_random_operands": {"registers_TFLOPs": 1635.0, "lds_fed_with_epilogue_TFLOPs": 1396.0,
                                                   "frac_of_lds_fed": issued * flops / gemm_seconds / 1396e12,
                                                   "source": "profiles/r03_mfma_peak_f16.jsonl (not measured by this run)"}}

    return {
        # `value`: the batch over the f32 rows as BASELINE c3 words it (the corpus adopted as it is).  `operand_image`: the same
        # batch once the corpus has its fp16 operand image -- a derived copy in MFMA operand order the library keeps for the corpora
        # it owns (the workspace store's); nominations only, every returned distance is re-scored in f64 from the f32 rows, the
        # answers are identical.
        "metric": "queries/sec at 10M-chunk corpus (batched)", "value": nq / wall_f32, "unit": "queries/s",
        "ms_per_batch": wall_f32 * 1e3, "rows_scanned_per_s": nq * rows / wall_f32,
        "config": {"workload": f"c3: {nq} batched queries x {rows} chunks (D=256, f32), top-{k}, one MI355X"},
        # The score matrix only nominates candidates.  On shards <= 32 M rows the library's default is f16 x 1 from 256 queries
        # (one fp16 operand per row and per query = 1 x the algorithmic flops on the 16-bit MFMA pipe, dense peak 2.5 PF), f16 x 2
        # from 128 (hi + lo per query: 2 x); smaller batches / larger shards use bf16 x 3 (3 x).  `frac` is ISSUED MFMA flops over
        # the 16-bit peak -- a mode that issues fewer MFMAs for the same answers finishes sooner at a LOWER fraction; queries/s is
        # the figure of merit.  The same batch on f32 MFMAs (roofline_f32_mfma) is bounded by 157.3 TF.
        "roofline": k3_roofline(gemm_s_f32, n_g_f32 // reps, "c3_f32_rows", ROW_BYTES, issued_f32),
        "operand_image": {"queries_per_s": nq / wall, "ms_per_batch": wall * 1e3, "build_ms": prepack_s * 1e3, "bytes": image_bytes,
                          "bytes_per_row": image_bytes / max(rows, 1), "answers_identical_to_f32_rows_run": f"{image_same}/{nq}",
                          "roofline": k3_roofline(gemm_s, n_g // reps, "c3", ROW_BYTES // 2, issued_img)},
        "single_query_same_corpus": single,
        "roofline_f32_mfma": f32_leg,
        "checks": {"torch_fp64_topk_match": ok, "k2_path_agreement": f"{n_same}/{nq}",
                   "selects_without_exactness_certificate": uncertain},
    }


def bench_c5(smt, ctx, device, rows, k, nq=1000, nlist=4096, nprobe=8, rerank=128):
    """Config c5 scaled to ONE GPU: IVF-PQ (nlist 4096, m 32) over a clustered corpus: build time, recall@k against
    the exact batched search, queries/s.  No reference semantics exist for this index.  The corpus has
    20 000 topics (NOT one per list) and the queries are independent draws from the generative model (NOT perturbed
    corpus rows); profiles/r02_ivf_sweep_*.json hold the nprobe x re-score-depth sweeps on this and a 500-topic corpus."""
    from tests import synth

    gen = synth.clustered_model_torch(20000, 8, 11, device)
    x = synth.clustered_sample_torch(gen, rows, 12)
    q = synth.clustered_sample_torch(gen, nq, 13).cpu().numpy()
    del gen
    torch.cuda.synchronize(device)
    corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
    exact = corpus.search(q, top_k=k)
    qd = torch.from_numpy(q).to(device)
    o_rows = torch.empty((nq, k), dtype=torch.int64, device=device)
    o_dist = torch.empty((nq, k), dtype=torch.float64, device=device)

    def one_coding(local_pca):
        t0 = time.perf_counter()
        ix = smt.IvfPq(corpus, nlist=nlist, train_iters=10, local_pca=local_pca)
        build_s = time.perf_counter() - t0
        info = ix.info()
        ix.search(q, top_k=k, nprobe=nprobe, rerank=rerank)  # warm-up (allocations)
        t0 = time.perf_counter()
        got = ix.search(q, top_k=k, nprobe=nprobe, rerank=rerank)
        dt_host = time.perf_counter() - t0        # host in, host out (pageable upload + result copies included)
        # device-resident form, like the other legs: queries and results stay in HBM, 5 batches back to back
        ix.search_device(qd.data_ptr(), nq, k, nprobe, rerank, 0, o_rows.data_ptr(), o_dist.data_ptr())
        torch.cuda.synchronize(device)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ix.search_device(qd.data_ptr(), nq, k, nprobe, rerank, 0, o_rows.data_ptr(), o_dist.data_ptr())
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / reps          # (whole calls, no profiling events inside)
        ctx.prof_enable(True)
        ctx.prof_reset()
        for _ in range(reps):
            ix.search_device(qd.data_ptr(), nq, k, nprobe, rerank, 0, o_rows.data_ptr(), o_dist.data_ptr())
        torch.cuda.synchronize(device)
        n_adc, ms_adc = ctx.prof_read("ivf_adc")
        n_pr, ms_pr = ctx.prof_read("ivf_probe")
        ctx.prof_enable(False)
        dev_rows = o_rows.cpu().numpy().view(np.uint64)
        same = all(dev_rows[i, :len(got[i][0])].tolist() == got[i][0].tolist() for i in range(nq))
        hit = sum(len(set(r.tolist()) & set(e.tolist())) for (r, _), (e, _) in zip(got, exact))
        # Algorithmic bytes of one ADC launch: every probed list's codes once (32 B code + 4 B row id per row,
        # nprobe / nlist of the corpus per query) + the re-scored rows (1 KiB each: `rerank` per probed list and query).
        sizes = ix.list_sizes().astype(np.float64)
        mean_probed = float((sizes * sizes).sum() / max(sizes.sum(), 1.0))   # a probed list is hit in proportion to its size
        code_bytes = nq * nprobe * mean_probed * 36.0
        rescore_bytes = nq * nprobe * min(rerank, mean_probed) * 1024.0
        adc_s = ms_adc / max(n_adc, 1) * 1e-3
        ix.close()
        return {"coding": "per-list PCA basis + 8-bit scalar quantisers (32 B/row; what the workspace store builds)" if local_pca
                          else "product quantisation, m = 32 sub-quantisers x 256 codes on coarse residuals (BASELINE c5's 'PQ m=32')",
                "build_s": build_s, "build_ms": info["build_ms"], "build_ms_stages": ["coarse k-means", "assign all rows", "quantiser training", "sort + encode"],
                "index_bytes": info["index_bytes"], "recall_at_k_vs_exact": hit / (nq * k), "queries_per_s": nq / dt, "ms_per_batch": dt * 1e3,
                "host_call_queries_per_s": nq / dt_host, "device_and_host_forms_agree": bool(same),
                "roofline": {"kernel": "ivf_adc_kernel (ADC scan of the probed lists + in-kernel re-score of the shortlist)", "bound": "hbm",
                             "achieved": (code_bytes + rescore_bytes) / adc_s / 1e9 if n_adc else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": (code_bytes + rescore_bytes) / adc_s / 1e9 / HBM_PEAK_GBPS if n_adc else None,
                             "traffic": measured_traffic("ivf_adc_lpca" if local_pca else "ivf_adc_pq", rows)[0],
                             "traffic_source": measured_traffic("ivf_adc_lpca" if local_pca else "ivf_adc_pq", rows)[1],
                             "algorithmic_bytes_per_launch": code_bytes + rescore_bytes, "code_bytes": code_bytes, "rescored_row_bytes": rescore_bytes,
                             "adc_ms_per_batch": adc_s * 1e3, "probe_ms_per_batch": ms_pr / max(n_pr, 1), "launches": n_adc,
                             "valu_busy_frac": (_traffic_entry("ivf_adc_lpca" if local_pca else "ivf_adc_pq") or {}).get("valu_busy_frac"),
                             "note": "counters (profiles/r06_ivf/, round 6 binary): per-list PCA codes: measured HBM traffic 2.19 GB per launch at 6.8 TB/s under "
                                     "the profiler (0.85 of peak: the part's read ceiling), VALUs 0.65 busy, no LDS conflicts; global PQ: 2.23 GB at 5.5 TB/s "
                                     "(2.41 GB before the XCD-aware block order), LDS bank conflicts 55 % of its LDS cycles in round 5 -> 0 (conflict-free LUT walk)"}}

    shipped = one_coding(True)
    try:
        pq = one_coding(False)
    except Exception as exc:
        pq = {"error": repr(exc)}
    corpus.close()
    del x
    torch.cuda.empty_cache()
    out = {"config": {"workload": f"c5 on one GPU: IVF index nlist={nlist}, 32 B codes per row, over {rows} chunks in 20000 "
                                  f"topics, {nq} independent queries, nprobe={nprobe}, {rerank} ADC candidates per list re-scored, "
                                  f"top-{k}; two codings: per-list PCA (shipped) and global PQ m=32 (as BASELINE c5 names it)"}}
    out.update(shipped)                       # the shipped coding's figures at the top level (as in rounds 1-2)
    out["global_pq_m32"] = pq
    # The 20 000-topic corpus above is EASY for probing (a topic sits inside one list: recall is the same at nprobe 1 and 128), so it
    # says nothing about the nprobe trade-off SURVEY 8(d) asks for.  The same generative model with 64 broad topics of 32 latent
    # dimensions (spread 0.8) puts ~64 lists on every topic and a query's true neighbours into many of them: recall RISES with nprobe
    # (tools/probe_ivf_corpus.py, profiles/r06_probe_ivf_corpus.jsonl: 0.23 / 0.73 / 0.98 / 0.997 at nprobe 1 / 8 / 32 / 128; the 500-topic
    # corpus of the round-2 sweeps is flat from nprobe 8 on).  Both codings, nprobe 8 / 32 / 128.
    try:
        gen = synth.clustered_model_torch(64, 32, 21, device)
        x = synth.clustered_sample_torch(gen, rows, 22, spread=0.8)
        qh = synth.clustered_sample_torch(gen, nq, 23, spread=0.8).cpu().numpy()
        del gen
        torch.cuda.synchronize(device)
        hard = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
        hard_exact = hard.search(qh, top_k=k)
        out["hard_corpus"] = {"config": {"workload": f"the same index over {rows} chunks in 64 broad topics (32 latent dimensions, spread 0.8: ~{nlist // 64} "
                                                     f"lists per topic): recall@{k} and queries/s by nprobe, both codings, {rerank} ADC candidates per list re-scored"}}
        for name, lp in (("per_list_pca", True), ("global_pq_m32", False)):
            out["hard_corpus"][name] = _c5_probe_sweep(smt, ctx, device, hard, qh, hard_exact, k, lp, nlist, rerank, (8, 32, 128))
        hard.close()
        del x
        torch.cuda.empty_cache()
    except Exception as exc:
        out["hard_corpus"] = {"error": repr(exc)}
    return out


def _c5_probe_sweep(smt, ctx, device, corpus, q, exact, k, local_pca, nlist, rerank, nprobes):
    """One index over `corpus` (coding by local_pca), searched with every nprobe: recall@k against `exact`, queries/s of the
    device-resident form (5 batches back to back), agreement of the host and device forms."""
    nq = len(q)
    qd = torch.from_numpy(q).to(device)
    o_rows = torch.empty((nq, k), dtype=torch.int64, device=device)
    o_dist = torch.empty((nq, k), dtype=torch.float64, device=device)
    t0 = time.perf_counter()
    ix = smt.IvfPq(corpus, nlist=nlist, train_iters=10, local_pca=local_pca)
    res = {"build_s": time.perf_counter() - t0, "index_bytes": ix.info()["index_bytes"]}
    for nprobe in nprobes:
        got = ix.search(q, top_k=k, nprobe=nprobe, rerank=rerank)
        hit = sum(len(set(r.tolist()) & set(e.tolist())) for (r, _), (e, _) in zip(got, exact))
        ix.search_device(qd.data_ptr(), nq, k, nprobe, rerank, 0, o_rows.data_ptr(), o_dist.data_ptr())
        torch.cuda.synchronize(device)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ix.search_device(qd.data_ptr(), nq, k, nprobe, rerank, 0, o_rows.data_ptr(), o_dist.data_ptr())
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / reps
        dev_rows = o_rows.cpu().numpy().view(np.uint64)
        same = all(dev_rows[i, :len(got[i][0])].tolist() == got[i][0].tolist() for i in range(nq))
        res[f"nprobe_{nprobe}"] = {"recall_at_k_vs_exact": hit / (nq * k), "queries_per_s": nq / dt, "ms_per_batch": dt * 1e3,
                                   "checks": {"device_and_host_forms_agree": bool(same)}}
    ix.close()
    return res


def bench_c5_full(smt, ctx, device, rows, k, nq=1000, nlist=4096, rerank=128):
    """BASELINE config c5 at its NAMED size on one GPU: 100 M chunks (102 GB of rows resident), nlist 4096, 32-byte codes (the shipped
    per-list PCA coding), build + query; recall@k against the exact batched search over the same rows.  (On eight GPUs every rank
    indexes 12.5 M rows: the c5 leg above, at 10 M rows, is about one rank's share.)"""
    from tests import synth

    free_b, _ = torch.cuda.mem_get_info(device)
    need = rows * 1024 * 1.12 + (8 << 30)
    if free_b < need:
        return {"skipped": f"needs {need / 1e9:.0f} GB of free HBM, {free_b / 1e9:.0f} GB are free"}
    gen = synth.clustered_model_torch(20000, 8, 11, device)
    x = synth.clustered_sample_torch(gen, rows, 12)
    q = synth.clustered_sample_torch(gen, nq, 13).cpu().numpy()
    del gen
    torch.cuda.synchronize(device)
    corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
    t0 = time.perf_counter()
    exact = corpus.search(q, top_k=k)
    exact_s = time.perf_counter() - t0
    qd = torch.from_numpy(q).to(device)
    o_rows = torch.empty((nq, k), dtype=torch.int64, device=device)
    o_dist = torch.empty((nq, k), dtype=torch.float64, device=device)
    t0 = time.perf_counter()
    ix = smt.IvfPq(corpus, nlist=nlist, train_iters=10, local_pca=True)
    build_s = time.perf_counter() - t0
    info = ix.info()
    out = {"config": {"workload": f"c5 at its named size on one GPU: IVF index nlist={nlist}, 32 B codes per row, over {rows} chunks in 20000 topics, "
                                  f"{nq} independent queries, {rerank} ADC candidates per list segment re-scored, top-{k}"},
           "rows": rows, "build_s": build_s, "build_ms": info["build_ms"], "index_bytes": info["index_bytes"], "exact_batch_search_s": exact_s}
    for nprobe in (8, 1, 32, 128):     # (SURVEY 8(d) c5 names nprobe 8 / 32 / 128)
        got = ix.search(q, top_k=k, nprobe=nprobe, rerank=rerank)
        hit = sum(len(set(r.tolist()) & set(e.tolist())) for (r, _), (e, _) in zip(got, exact))
        ix.search_device(qd.data_ptr(), nq, k, nprobe, rerank, 0, o_rows.data_ptr(), o_dist.data_ptr())
        torch.cuda.synchronize(device)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ix.search_device(qd.data_ptr(), nq, k, nprobe, rerank, 0, o_rows.data_ptr(), o_dist.data_ptr())
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / reps
        dev_rows = o_rows.cpu().numpy().view(np.uint64)
        same = all(dev_rows[i, :len(got[i][0])].tolist() == got[i][0].tolist() for i in range(nq))
        out[f"nprobe_{nprobe}"] = {"recall_at_k_vs_exact": hit / (nq * k), "queries_per_s": nq / dt, "ms_per_batch": dt * 1e3,
                                   "checks": {"device_and_host_forms_agree": bool(same)}}
    # more probes must never cost recall (the segments of a probed list share out its codes evenly whatever nprobe: ivfpq_search.hip)
    rc = [out[f"nprobe_{p}"]["recall_at_k_vs_exact"] for p in (8, 32, 128)]
    out["checks"] = {"recall_does_not_fall_with_nprobe": bool(rc[1] >= rc[0] - 0.002 and rc[2] >= rc[1] - 0.002)}
    ix.close()
    # ... and the coding BASELINE config 5 NAMES -- "PQ m=32" -- at the same size: global product quantisation, m = 32 x 256 codes
    try:
        out["global_pq_m32"] = _c5_probe_sweep(smt, ctx, device, corpus, q, exact, k, False, nlist, rerank, (8, 32))
    except Exception as exc:
        out["global_pq_m32"] = {"error": repr(exc)}
    corpus.close()
    del x
    torch.cuda.empty_cache()
    return out


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


if __name__ == "__main__":
    sys.exit(main())
