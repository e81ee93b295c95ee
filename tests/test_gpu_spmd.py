"""GPU: the ONE-RANK-PER-PROCESS group with n_ranks in {2, 4} -- the process model of a torchrun / MPI driver (what bench.py --gpus N
runs under torch.distributed.run) -- executed on one GPU against a test double for RCCL (tests/fake_rccl, mapped before the library
looks for librccl.so.1: no product change).

What runs here that no logical-shard test reaches (semtools_amd/csrc/group.cpp): smt_group_create_rank -> ncclCommInitRank with
n > 1, group_barrier, the packed k-list ncclAllGather + merge on every rank, the per-rank status words that make all ranks return
an error together, group_agree, exchange_host_lists' count + padded all-gathers (threshold mode, k > 56), the shared-centroid
ncclAllReduce, the per-rank streaming of save / load, dealt appends and the sharded embed with ranks in different processes.
RCCL itself and xGMI stay unmeasured.  Contract (include/semtools_hip.h): sharded result == smt_search on the
unsharded matrix, which is also compared with the oracle here (north_star: the exchange behind src/workspace/store.rs:481-546 and
src/search/mod.rs:84-119 once the corpus is sharded)."""
import glob
import os
import pickle
import subprocess
import sys
import time

import numpy as np
import pytest

from tests import synth
from tests import spmd_worker as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spawn(world, tmp_path, scenarios, timeout=420):
    from tests import fake_rccl

    fake = fake_rccl.build()
    tag = f"t{os.getpid()}"
    env = dict(os.environ, SEMTOOLS_NO_TORCH_PRELOAD="1", FAKE_RCCL_TIMEOUT_S="30", FAKE_RCCL_TAG=tag, PYTHONPATH=ROOT)
    env.pop("SEMTOOLS_GROUP_HOST_LISTS", None)
    procs = []
    try:
        for r in range(world):
            log = open(tmp_path / f"rank{r}.log", "wb")
            procs.append((subprocess.Popen([sys.executable, "-m", "tests.spmd_worker", "--rank", str(r), "--world", str(world), "--dir", str(tmp_path),
                                            "--fake", fake, "--scenarios", scenarios], cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT), log))
        t0 = time.time()
        for p, _ in procs:
            try:
                p.wait(timeout=max(1.0, timeout - (time.time() - t0)))
            except subprocess.TimeoutExpired:
                pass
        hung = [i for i, (p, _) in enumerate(procs) if p.poll() is None]
    finally:
        for p, log in procs:
            if p.poll() is None:
                p.kill()          # (the exact processes started above)
                p.wait()
            log.close()
        for f in glob.glob(f"/dev/shm/fake_rccl_{tag}_*"):
            os.unlink(f)
    logs = "".join(f"\n--- rank {r}:\n" + (tmp_path / f"rank{r}.log").read_text(errors="replace")[-3000:] for r in range(world))
    assert not hung, f"ranks {hung} were still running after {timeout} s (a rank left inside a collective?){logs}"
    assert all(p.returncode == 0 for p, _ in procs), f"exit codes {[p.returncode for p, _ in procs]}{logs}"
    outs = [pickle.load(open(tmp_path / f"out_{r}.pkl", "rb")) for r in range(world)]
    for o in outs:
        assert not o["errors"], "rank %d:\n%s" % (o["rank"], "\n".join(f"[{k}]\n{v}" for k, v in o["errors"].items()))
    return outs


def _lists(res):
    return [(r.tolist(), d.tolist()) for r, d in res]


@pytest.fixture(scope="module")
def plain(gpu_ctx):
    import semtools_amd as smt

    emb = synth.unit_rows(W.N_ROWS, seed=3)
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    yield emb, c
    c.close()


def _check_group(outs, world):
    per = -(-W.N_ROWS // world)
    for r, o in enumerate(outs):
        assert o["info"] == dict(n_ranks=world, n_local=1, first_rank=r, rccl_ranks=world, rccl_version=22099), o["info"]   # the double's version
        assert o["transport"] == "rccl"                       # ranks in different processes: the all-gather is the only transport
        assert o["rank_rows"] == [max(0, min(per, W.N_ROWS - i * per)) for i in range(world)]
        assert o["fake_stats"]["allgathers"] > 0 and o["fake_stats"]["chunks"] >= o["fake_stats"]["allgathers"]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_every_search_mode_with_one_rank_per_process(plain, tmp_path, world):
    from oracle import oracle as orc

    emb, c = plain
    outs = _spawn(world, tmp_path, "modes,pipelined")
    _check_group(outs, world)
    qs = synth.unit_query(4, nq=3)
    want = [_lists(c.search(qs, **kw)) for kw in W.CASES]
    assert len(want[2][0][0]) > 7 and len(want[5][0][0]) == 2000 and want[9][0][0] == []   # the cases are what their comments say
    # the unsharded answers themselves against the oracle: rows bit-exact, f64 distances equal to the accurate form
    for kw, per_q in zip(W.CASES, want):
        if "ranges" in kw or kw.get("mode"):
            continue
        for q, (rows, dist) in zip(qs, per_q):
            ref = orc.search_documents(emb, [len(emb)], q, n_lines=0, top_k=kw["top_k"], max_distance=kw.get("max_distance"), accurate=True)
            assert rows == [r["match_line"] for r in ref] and dist == [r["distance"] for r in ref]
    q12, q140 = synth.unit_query(5, nq=12), synth.unit_query(6, nq=140)
    want12, want140, want_thr5 = _lists(c.search(q12, top_k=10)), _lists(c.search(q140, top_k=7)), _lists(c.search(q140[:5], max_distance=0.88))
    q16 = synth.unit_query(21, nq=16)
    want16 = c.search(q16, top_k=W.K_DEV)
    from tests.test_gpu_nearties import adversarial_corpus, _oracle_topk

    qa, emb_a, _ = adversarial_corpus(n=6000, n_cluster=50 * world, seed=57)
    q5 = synth.unit_query(95, nq=5)
    q5[3] = qa
    for o in outs:                                            # EVERY rank holds the whole answer
        p = o["pipelined"]
        assert p["verdicts"] == [0, 0, 0, 1, 0], (o["rank"], p["verdicts"])      # smt_sharded_search_topk_device_ex: worst status of the shards
        for i in (0, 1, 2, 4):
            exp_rows, exp_dist = _oracle_topk(emb_a, q5[i], W.K_DEV)
            assert p["verdict_rows"][i].tolist() == exp_rows and np.array_equal(p["verdict_dist"][i], exp_dist), (o["rank"], i)
        for kw, got, w in zip(W.CASES, o["modes"], want):
            assert got == w, (o["rank"], kw)
        assert o["batch12"] == want12 and o["batch140"] == want140 and o["thr5"] == want_thr5, o["rank"]
        p = o["pipelined"]
        assert p["uncertain"] == 0
        for async_select in (0, 1):
            assert len(p[(async_select, "plan")]) == W.PIPELINED
            for s, (q0, nq) in enumerate(p[(async_select, "plan")]):
                for j in range(nq):
                    assert p[(async_select, "rows")][s, j].tolist() == want16[q0 + j][0].tolist(), (o["rank"], async_select, s)
                    assert np.array_equal(p[(async_select, "dist")][s, j], want16[q0 + j][1]), (o["rank"], async_select, s)
            for key, got in p.items():
                if key[0] == async_select and key[1] == "host":
                    q0, nq = p[(async_select, "plan")][key[2]]
                    assert got == _lists(want16[q0:q0 + nq])


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_store_side_and_index_build_with_one_rank_per_process(plain, tmp_path, gpu_ctx, world):
    import semtools_amd as smt
    from oracle import oracle as orc
    from tests.test_gpu_ivfpq import clustered, recall

    emb, c = plain
    outs = _spawn(world, tmp_path, "store,ivf")
    _check_group(outs, world)
    qs = synth.unit_query(4, nq=3)
    # ---- store: dealt appends -> pieces; the file written by n processes holds the rows in global order
    whole = np.concatenate([emb[:4001], synth.unit_rows(700, seed=77), synth.unit_rows(5, seed=78)])
    c2 = smt.Corpus(gpu_ctx)
    c2.append(whole)
    want = _lists(c2.search(qs, top_k=12))
    want_thr = _lists(c2.search(qs, top_k=5, max_distance=0.92))
    want_rng = _lists(c2.search(qs, top_k=5, ranges=[(3900, 4300), (4600, 4706)]))
    c_file = smt.Corpus.load(gpu_ctx, tmp_path / "sharded.f32")
    assert np.array_equal(c_file.read_rows(0, len(whole)), whole)
    c_file.close(); c2.close()
    table = synth.table(3000, seed=2)
    ids, offsets = synth.token_lines(900, V=3000, seed=9, min_tok=0, max_tok=20)
    ref_emb = orc.embed_lines(table, ids, offsets, True, 2048)
    c4 = smt.Corpus(gpu_ctx)
    c4.append(ref_emb)
    want_embed = _lists(c4.search(qs, top_k=8))
    c4.close()
    per = -(-len(whole) // world)
    for o in outs:
        s = o["store"]
        assert s["first"] == 4001 and s["first2"] == 4701
        assert sum(n for n, _ in s["layout"]) == len(whole) and sum(s["rank_rows"]) == len(whole)
        assert len(s["layout"]) > world                       # really a list of pieces, not one range per rank
        assert max(s["rank_rows"]) - min(s["rank_rows"]) <= 6    # the deal keeps the shards level
        assert s["search"] == want and s["search_thr"] == want_thr and s["search_rng"] == want_rng, o["rank"]
        assert s["loaded_rank_rows"] == [max(0, min(per, len(whole) - i * per)) for i in range(world)]
        assert s["loaded_search"] == want
        assert s["embed_first"] == 0 and sum(s["embed_rank_rows"]) == 900
        at = 0
        for n, r in s["embed_layout"]:                        # block r was pooled on rank r: bit-exact against the oracle's pool
            if r == o["rank"]:
                assert np.array_equal(s["embed_host"][at:at + n], ref_emb[at:at + n]), (o["rank"], at)
            at += n
        assert s["embed_search"] == want_embed
    # ---- index: the all-reduce ran inside the k-means loop; shared centroids make a list mean the same thing on every shard
    x, _ = clustered(40000, 200, seed=12)
    q, _ = clustered(24, 200, seed=12)
    cx = smt.Corpus(gpu_ctx)
    cx.append(x)
    exact = cx.search(q, top_k=10)
    cx.close()
    for o in outs:
        v = o["ivf"]
        assert v["allreduces"] >= 2 * (1 + 5)                 # (sums + counts) x (seeding + 5 iterations)
        assert v["search"] == outs[0]["ivf"]["search"] and v["indep_search"] == outs[0]["ivf"]["indep_search"]
        for key in ("search", "indep_search"):
            got = [(np.asarray(r, dtype=np.uint64), np.asarray(d)) for r, d in v[key]]
            assert recall(got, exact) >= 0.95, (key, recall(got, exact))
            for (rows, dist), qq in zip(got, q):
                assert (np.diff(dist) >= 0).all() and len(set(rows.tolist())) == len(rows)
                ref = 1.0 - x[rows.astype(np.int64)].astype(np.float64) @ qq.astype(np.float64)
                assert np.allclose(dist, ref, atol=1e-6)
    s0, s1 = np.asarray(outs[0]["ivf"]["list_sizes"], float), np.asarray(outs[1]["ivf"]["list_sizes"], float)
    i0, i1 = np.asarray(outs[0]["ivf"]["indep_list_sizes"], float), np.asarray(outs[1]["ivf"]["indep_list_sizes"], float)
    assert s0.sum() == 40000 // world
    assert np.corrcoef(s0, s1)[0, 1] > 0.8 > abs(np.corrcoef(i0, i1)[0, 1])


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_a_rank_that_fails_locally_takes_every_rank_with_it(plain, tmp_path, world):
    """One rank's local stage fails (injected: smt_debug_group_fail_next).  Every rank must return THAT status from the same call --
    through the status word of the packed exchange, through group_agree, through the build's agreement -- none may be left inside a
    collective, and the group must still answer the next search."""
    from semtools_amd import _lib as L

    emb, c = plain
    outs = _spawn(world, tmp_path, "failures")
    _check_group(outs, world)
    qs = synth.unit_query(4, nq=3)
    want = _lists(c.search(qs, top_k=5))
    for o in outs:
        f = o["failures"]
        assert f[("stage", "code")] == L.SMT_E_NOMEM and f[("threshold", "code")] == L.SMT_E_IO, f
        assert f[("build", "code")] == L.SMT_E_NOMEM and f[("append", "code")] == L.SMT_E_INVALID, f
        assert f["append_rolled_back"], "a failed dealt append must leave every shard as it was"
        # a row outside the library's domain in ONE rank's share of a dealt append: refused there, failed and rolled back everywhere
        assert f[("domain", "code")] == L.SMT_E_INVALID and f["domain_rolled_back"], f
        assert f[("domain", "seconds")] < 20 and f[("domain", "after")] == want
        if o["rank"] == world - 1:
            assert "outside the library's domain" in f[("domain", "msg")], f[("domain", "msg")]
        for label in ("stage", "threshold", "build", "append"):
            assert f[(label, "seconds")] < 20, "a rank waited for the double's timeout: it was left inside a collective"
            assert f[(label, "after")] == want, (o["rank"], label)
            if o["rank"] == world - 1:
                assert "injected failure" in f[(label, "msg")], f[(label, "msg")]
            else:
                assert f"rank {world - 1} failed" in f[(label, "msg")] or "another shard" in f[(label, "msg")], f[(label, "msg")]
