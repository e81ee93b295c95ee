"""ONE RANK of an n-rank, one-rank-per-process smt_group, all ranks on GPU 0 -- TEST INFRASTRUCTURE (tests/test_gpu_spmd.py spawns n of these).

This is the process model a torchrun / MPI driver uses (semtools_amd/csrc/group.cpp: smt_group_unique_id on rank 0, the 128 bytes
travel by any means -- a file here --, smt_group_create_rank -> ncclCommInitRank, then SPMD calls).  RCCL refuses two ranks on one
device, so the communicator is tests/fake_rccl's double, mapped into the process BEFORE the library looks for "librccl.so.1"
(load_rccl tries RTLD_NOLOAD first: the product is unchanged).  Every rank runs the same scenarios on the same host arguments and
pickles what it got; the parent compares every rank's answers with the unsharded search and with the oracle.

No torch in this process: torch maps the real librccl.so.1 (same SONAME).  Device buffers come from the HIP runtime through ctypes."""
import argparse
import ctypes as C
import os
import pickle
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_ROWS = 6000
K_DEV = 10
PIPELINED = 100

CASES = [
    dict(top_k=7),
    dict(top_k=56),
    dict(top_k=7, max_distance=0.93),                                   # all rows under the threshold (A6): variable-length exchange
    dict(top_k=4, max_distance=0.95, mode=1),                           # workspace: score threshold, then top-k (A10)
    dict(top_k=100),                                                    # k > 56: host lists, count + padded all-gather
    dict(top_k=2000),
    dict(top_k=5, ranges=[(10, 50), (2999, 3001), (4000, 5999)]),       # path-subset filter crossing shard borders
    dict(top_k=3, ranges=[(5990, 6000)]),                               # a filter that leaves most shards nothing
    dict(top_k=9, max_distance=0.9, ranges=[(100, 4100)]),
    dict(top_k=3, max_distance=1e-9),                                   # nobody has a hit
    dict(top_k=6, max_distance=0.9, mode=1, ranges=[(0, 1700), (3100, 5000)]),
]


class Hip:
    """The few runtime calls a test needs for raw device buffers (libamdhip64 is already mapped by libsemtools_hip.so)."""

    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so.7")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipDeviceSynchronize.argtypes = []

    def _ok(self, e, what):
        if e != 0:
            raise RuntimeError(f"{what}: hipError {e}")

    def malloc(self, nbytes):
        p = C.c_void_p()
        self._ok(self.rt.hipMalloc(C.byref(p), max(int(nbytes), 8)), "hipMalloc")
        return int(p.value)

    def free(self, p):
        self.rt.hipFree(C.c_void_p(p))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        self._ok(self.rt.hipMemcpy(C.c_void_p(p), arr.ctypes.data_as(C.c_void_p), arr.nbytes, 1), "hipMemcpy H2D")
        return p

    def download(self, p, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self._ok(self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(p), out.nbytes, 2), "hipMemcpy D2H")
        return out

    def zero(self, p, nbytes):
        self._ok(self.rt.hipMemset(C.c_void_p(p), 0, int(nbytes)), "hipMemset")

    def sync(self):
        self._ok(self.rt.hipDeviceSynchronize(), "hipDeviceSynchronize")


def _lists(res):
    return [(r.tolist(), d.tolist()) for r, d in res]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--dir", required=True)
    ap.add_argument("--fake", required=True)
    ap.add_argument("--scenarios", default="modes,pipelined,store,ivf,failures")
    a = ap.parse_args()
    rank, world = a.rank, a.world
    out = {"rank": rank, "errors": {}}
    os.environ["SEMTOOLS_NO_TORCH_PRELOAD"] = "1"
    fake = C.CDLL(a.fake, mode=C.RTLD_GLOBAL)        # first: the library's dlopen("librccl.so.1", RTLD_NOLOAD) now finds THIS object
    import semtools_amd as smt
    from semtools_amd import _lib as L
    from tests import synth

    assert "torch" not in sys.modules
    uid_path = os.path.join(a.dir, "uid.bin")
    if rank == 0:
        uid = smt.Group.unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 120:
                raise RuntimeError("rank 0 never published the unique id")
            time.sleep(0.01)
        uid = open(uid_path, "rb").read()
    g = smt.Group.from_rank(0, rank, world, uid)
    out["info"] = g.info()
    out["transport"] = g.transport
    hip = Hip()
    emb = synth.unit_rows(N_ROWS, seed=3)
    sc = smt.ShardedCorpus(g, rows=emb)
    out["rank_rows"] = sc.rank_rows().tolist()
    qs = synth.unit_query(4, nq=3)
    scenarios = a.scenarios.split(",")

    def run(name, fn):
        if name not in scenarios:
            return
        try:
            fn()
        except Exception:
            out["errors"][name] = traceback.format_exc()

    # ---------------------------------------------------------------- every search mode through the exchange
    def modes():
        out["modes"] = [_lists(sc.search(qs, **kw)) for kw in CASES]
        out["batch12"] = _lists(sc.search(synth.unit_query(5, nq=12), top_k=10))          # >= 8 queries: the MFMA path inside the shard
        ql = synth.unit_query(6, nq=140)
        out["batch140"] = _lists(sc.search(ql, top_k=7))
        out["thr5"] = _lists(sc.search(ql[:5], max_distance=0.88))                        # several threshold queries per shard
        g.barrier()

    # ---------------------------------------------------------------- the device form, nothing synchronises between 100 calls
    def pipelined():
        q16 = synth.unit_query(21, nq=16)
        qd = hip.upload(q16)
        res = {}
        for async_select in (0, 1):
            g.ctx(0).set_tuning("async_select", async_select)
            outs = hip.malloc(PIPELINED * 3 * 2 * K_DEV * 8)
            hip.zero(outs, PIPELINED * 3 * 2 * K_DEV * 8)
            hip.sync()
            plan = []
            for s in range(PIPELINED):
                nq = 1 if s % 3 else 3
                q0 = s % (16 - nq + 1)
                if s % 40 == 39:
                    res[(async_select, "host", s)] = _lists(sc.search(q16[q0:q0 + nq], top_k=K_DEV))   # host form in between (synchronises)
                sc.search_topk_device([qd + q0 * 256 * 4], nq, K_DEV, [outs + s * 3 * 2 * K_DEV * 8])
                plan.append((q0, nq))
            g.synchronize()
            m = hip.download(outs, (PIPELINED, 3, 2, K_DEV), np.uint64)
            res[(async_select, "plan")] = plan
            res[(async_select, "rows")] = m[:, :, 0, :].copy()
            res[(async_select, "dist")] = m[:, :, 1, :].copy().view(np.float64)
            hip.free(outs)
        g.ctx(0).set_tuning("async_select", 0)
        res["uncertain"] = g.ctx(0).uncertain_count()
        hip.free(qd)
        # per-query verdicts (_ex): the ranks' status words ride in the same all-gather as their k-lists; worst status wins
        from tests.test_gpu_nearties import adversarial_corpus

        qa, emb_a, _ = adversarial_corpus(n=6000, n_cluster=50 * world, seed=57)
        q5 = synth.unit_query(95, nq=5)
        q5[3] = qa
        sca = smt.ShardedCorpus(g, rows=emb_a)
        qd5 = hip.upload(q5)
        o5 = hip.malloc(5 * 2 * K_DEV * 8)
        st5 = hip.upload(np.full(5, 7, dtype=np.uint32))
        hip.sync()
        sca.search_topk_device([qd5], 5, K_DEV, [o5], [st5])
        g.synchronize()
        res["verdicts"] = hip.download(st5, (5,), np.uint32).tolist()
        m5 = hip.download(o5, (5, 2, K_DEV), np.uint64)
        res["verdict_rows"] = m5[:, 0, :].copy()
        res["verdict_dist"] = m5[:, 1, :].copy().view(np.float64)
        res["verdict_uncertain_local"] = g.ctx(0).uncertain_count()
        for p in (qd5, o5, st5):
            hip.free(p)
        sca.close()
        out["pipelined"] = res

    # ---------------------------------------------------------------- the store's side: dealt appends, file round trip, sharded embed
    def store():
        res = {}
        sc2 = smt.ShardedCorpus(g, rows=emb[:4001])
        extra = synth.unit_rows(700, seed=77)
        res["first"] = sc2.append(extra)                       # dealt over the ranks: the numbering becomes a list of pieces
        res["first2"] = sc2.append(synth.unit_rows(5, seed=78))   # a handful of rows goes to one shard
        res["layout"] = sc2.layout()
        res["rank_rows"] = sc2.rank_rows().tolist()
        res["search"] = _lists(sc2.search(qs, top_k=12))
        res["search_thr"] = _lists(sc2.search(qs, top_k=5, max_distance=0.92))
        res["search_rng"] = _lists(sc2.search(qs, top_k=5, ranges=[(3900, 4300), (4600, 4706)]))
        path = os.path.join(a.dir, "sharded.f32")
        sc2.save(path)                                         # rank 0 makes the file, every rank streams its own pieces
        g.barrier()
        sc3 = smt.ShardedCorpus.load(g, path)                  # cut into ceil(N / n) ranges again
        res["loaded_rank_rows"] = sc3.rank_rows().tolist()
        res["loaded_search"] = _lists(sc3.search(qs, top_k=12))
        sc3.close()
        # sharded embed: the lines are dealt to the ranks in blocks, block r pooled on rank r, appended to rank r's shard
        table = synth.table(3000, seed=2)
        ids, offsets = synth.token_lines(900, V=3000, seed=9, min_tok=0, max_tok=20)
        sm = smt.ShardedModel(g, table, normalize=True)
        sc4 = smt.ShardedCorpus(g, empty=True)
        host, first = sm.embed(ids, offsets, max_tokens=2048, append_to=sc4)
        res["embed_first"] = first
        res["embed_rank_rows"] = sc4.rank_rows().tolist()
        res["embed_layout"] = sc4.layout()
        res["embed_host"] = host        # (a multi-process group fills in the blocks of the local ranks only)
        res["embed_search"] = _lists(sc4.search(qs, top_k=8))
        sm.close(); sc4.close(); sc2.close()
        out["store"] = res

    # ---------------------------------------------------------------- shared-centroid index build: ncclAllReduce inside the k-means loop
    def ivf():
        from tests.test_gpu_ivfpq import clustered

        x, _ = clustered(40000, 200, seed=12)
        q, _ = clustered(24, 200, seed=12)
        scx = smt.ShardedCorpus(g, rows=x)
        res = {}
        before = C.c_uint64()
        fake.fake_rccl_stats(None, C.byref(before), None, None)
        shared = smt.ShardedIvfPq(scx, nlist=64, train_iters=5, shared_centroids=True)
        after = C.c_uint64()
        fake.fake_rccl_stats(None, C.byref(after), None, None)
        res["allreduces"] = int(after.value - before.value)
        res["list_sizes"] = shared.shard_list_sizes(0, 64).tolist()
        res["search"] = _lists(shared.search(q, top_k=10, nprobe=16, rerank=128))
        indep = smt.ShardedIvfPq(scx, nlist=64, train_iters=5, shared_centroids=False)
        res["indep_list_sizes"] = indep.shard_list_sizes(0, 64).tolist()
        res["indep_search"] = _lists(indep.search(q, top_k=10, nprobe=16, rerank=128))
        shared.close(); indep.close(); scx.close()
        out["ivf"] = res

    # ---------------------------------------------------------------- one rank fails locally: all return the same error, nobody hangs
    def failures():
        res = {}
        victim = world - 1

        def attempt(label, kind, code, call):
            if rank == victim:
                g.debug_fail_next(kind, code)
            t0 = time.time()
            try:
                call()
                res[(label, "code")] = 0
            except smt.SmtError as e:
                res[(label, "code")] = e.code
                res[(label, "msg")] = str(e)
            res[(label, "seconds")] = time.time() - t0
            res[(label, "after")] = _lists(sc.search(qs, top_k=5))      # the group is still usable, and in step

        attempt("stage", 1, L.SMT_E_NOMEM, lambda: sc.search(qs, top_k=5))                          # SMT_DEBUG_FAIL_STAGE
        attempt("threshold", 2, L.SMT_E_IO, lambda: sc.search(qs, top_k=5, max_distance=0.9))      # SMT_DEBUG_FAIL_AGREE (host-list path)
        x = synth.unit_rows(4000, seed=31)
        scx = smt.ShardedCorpus(g, rows=x)
        attempt("build", 3, L.SMT_E_NOMEM, lambda: smt.ShardedIvfPq(scx, nlist=32, train_iters=3, shared_centroids=True).close())   # _BUILD
        # the agreement again through another caller: a dealt append must leave every shard as it was
        rows_before = scx.rank_rows().tolist()
        attempt("append", 2, L.SMT_E_INVALID, lambda: scx.append(synth.unit_rows(300, seed=5)))
        res["append_rolled_back"] = scx.rank_rows().tolist() == rows_before and scx.rows == 4000
        # ... and a failure nobody injected: ONE row of a dealt append is outside the library's domain (a NaN component): it lands in
        # one rank's share, that rank refuses it (domain.hip), and the append must fail on EVERY rank and roll back everywhere
        bad = synth.unit_rows(400, seed=6)
        bad[399, 17] = np.nan                                   # the last row: the last rank's share
        t0 = time.time()
        try:
            scx.append(bad)
            res[("domain", "code")] = 0
        except smt.SmtError as e:
            res[("domain", "code")] = e.code
            res[("domain", "msg")] = str(e)
        res[("domain", "seconds")] = time.time() - t0
        res["domain_rolled_back"] = scx.rank_rows().tolist() == rows_before and scx.rows == 4000
        res[("domain", "after")] = _lists(sc.search(qs, top_k=5))
        scx.close()
        out["failures"] = res

    run("modes", modes)
    run("pipelined", pipelined)
    run("store", store)
    run("ivf", ivf)
    run("failures", failures)

    st = [C.c_uint64() for _ in range(4)]
    fake.fake_rccl_stats(*[C.byref(x) for x in st])
    out["fake_stats"] = dict(allgathers=int(st[0].value), allreduces=int(st[1].value), payload_bytes=int(st[2].value), chunks=int(st[3].value))
    try:
        g.barrier()
    except Exception:
        out["errors"]["final_barrier"] = traceback.format_exc()
    sc.close()
    g.close()
    with open(os.path.join(a.dir, f"out_{rank}.pkl.tmp"), "wb") as f:
        pickle.dump(out, f)
    os.rename(os.path.join(a.dir, f"out_{rank}.pkl.tmp"), os.path.join(a.dir, f"out_{rank}.pkl"))


if __name__ == "__main__":
    main()
