"""GPU: adversarial near-ties around the k-th place.

The f32 scan nominates k + 8 rows per list; more than 8 rows within f32 noise of the k-th distance used to be an
unproven assumption.  Now the select stage carries an exactness certificate and the host entry points re-answer a
query exhaustively when it fails.  Two contracts are pinned here:
  * vs the oracle's f64-accurate ordering (what the library promises): indices EXACT, distances bit-equal;
  * vs the serial-f32 restatement (the closest thing to what the reference executes): tie-aware.
Rows of the cluster are rescaled and 1-ulp-perturbed copies of one vector -- what "foo" vs "foo foo foo" or the
same tokens in another order produce in a real corpus."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import synth
from tests.compare import assert_topk_tie_aware, reference_distances

pytestmark = pytest.mark.gpu


def _unit(x):
    return (x / np.linalg.norm(x)).astype(np.float32)


def adversarial_corpus(n=4000, n_better=5, n_cluster=40, seed=21):
    rng = np.random.default_rng(seed)
    q = _unit(rng.standard_normal(256))
    emb = synth.unit_rows(n, seed=seed + 1, dup_frac=0.0, zero_frac=0.0)
    v = _unit(q + 0.9 * _unit(rng.standard_normal(256)))                 # the cluster's direction
    pos = rng.choice(n, size=n_better + n_cluster, replace=False)
    for i, p in enumerate(pos[:n_better]):                               # clearly better rows
        emb[p] = _unit(q + (0.3 + 0.05 * i) * _unit(rng.standard_normal(256)))
    for i, p in enumerate(pos[n_better:]):
        if i % 3 == 0:
            row = (v * np.float32(0.37 + 0.11 * i)).astype(np.float32)   # rescaled copy ("foo foo foo" vs "foo")
        elif i % 3 == 1:
            row = v.copy()                                               # 1-ulp perturbations (another summation order)
            idx = rng.choice(256, size=6, replace=False)
            row[idx] = np.nextafter(row[idx], np.float32(np.inf if i % 2 else -np.inf), dtype=np.float32)
        else:
            row = v.copy()                                               # exact duplicates: must come back in row order
        emb[p] = row
    return q, np.ascontiguousarray(emb), sorted(pos[n_better:].tolist())


def _oracle_topk(emb, q, k):
    res = orc.search_documents(emb, [len(emb)], q, n_lines=0, top_k=k, accurate=True)
    return [r["match_line"] for r in res], np.array([r["distance"] for r in res])


@pytest.mark.parametrize("k", [3, 10, 20, 44, 56, 60, 64, 100])
def test_cluster_around_the_kth_place_single_query(gpu_ctx, k):
    import semtools_amd as smt

    q, emb, cluster = adversarial_corpus()
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    rows, dist = c.search(q, top_k=k)[0]
    orows, odist = _oracle_topk(emb, q, k)
    assert rows.tolist() == orows                       # the promise: exact f64 ordering, ties by row
    assert np.array_equal(dist, odist)
    ref = reference_distances(emb, q, accurate=False)   # the reference's arithmetic (serial f32)
    assert_topk_tie_aware(rows, dist, ref, k)
    if 5 < k < 45:                                      # the k-th place lies inside the 40-row cluster
        assert set(rows[5:k].tolist()) <= set(cluster)
    c.close()


def test_cluster_batched_mfma_path_and_ranges(gpu_ctx):
    import semtools_amd as smt

    q, emb, cluster = adversarial_corpus(n=6000, seed=33)
    qs = np.concatenate([q[None], synth.unit_query(2, nq=11)])     # 12 queries -> K3; query 0 is the adversarial one
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    for k in (10, 30):
        got = c.search(qs, top_k=k)
        for i in range(len(qs)):
            orows, odist = _oracle_topk(emb, qs[i], k)
            assert got[i][0].tolist() == orows, (k, i)
            assert np.array_equal(got[i][1], odist)
            assert_topk_tie_aware(got[i][0], got[i][1], reference_distances(emb, qs[i]), k)
    # range-filtered (workspace path subset): K2 with the chunk table, same certificate
    ranges = [(0, 2500), (3000, 6000)]
    elig = [r for b, e in ranges for r in range(b, e)]
    rows, dist = c.search(q, top_k=10, ranges=ranges)[0]
    sub = emb[elig]
    res = orc.search_documents(sub, [len(sub)], q, n_lines=0, top_k=10, accurate=True)
    assert rows.tolist() == [elig[r["match_line"]] for r in res]
    assert_topk_tie_aware(rows, dist, reference_distances(emb, q), 10, rows_subset=elig)
    c.close()


def test_threshold_through_the_cluster(gpu_ctx):
    """max_distance inside the cluster: strict `<` on the exact f64 value (mod.rs:88-89); against the serial-f32
    restatement membership may differ only within 1e-5 of the threshold."""
    import semtools_amd as smt

    q, emb, cluster = adversarial_corpus(seed=44)
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    acc = reference_distances(emb, q, accurate=True)
    md = float(np.median(acc[cluster]))
    rows, dist = c.search(q, top_k=3, max_distance=md)[0]
    want = orc.search_documents(emb, [len(emb)], q, n_lines=0, top_k=3, max_distance=md, accurate=True)
    assert rows.tolist() == [r["match_line"] for r in want]
    assert np.array_equal(dist, [r["distance"] for r in want])
    ser = reference_distances(emb, q, accurate=False)
    got = set(rows.tolist())
    assert set(np.nonzero(ser < md - 1e-5)[0].tolist()) <= got
    assert not (got & set(np.nonzero(ser >= md + 1e-5)[0].tolist()))
    # workspace semantics: score threshold, then ALWAYS top-k (store.rs:502-503, :543)
    rows_w, dist_w = c.search(q, top_k=8, max_distance=md, mode=smt.MODE_WORKSPACE)[0]
    keep = [(d, r) for r, d in enumerate(acc) if (1.0 - d) > float(np.float32(1.0) - np.float32(md))]
    keep.sort()
    assert rows_w.tolist() == [r for _, r in keep[:8]]
    c.close()


def test_device_entry_point_counts_what_it_cannot_prove(gpu_ctx):
    import torch
    import semtools_amd as smt

    q, emb, _ = adversarial_corpus(seed=55)
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    gpu_ctx.uncertain_count()                                  # reset
    qd = torch.from_numpy(q).cuda()
    o_rows = torch.empty(10, dtype=torch.int64, device="cuda")
    o_dist = torch.empty(10, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    c.search_topk_device(qd.data_ptr(), 1, 10, 0, o_rows.data_ptr(), o_dist.data_ptr())
    gpu_ctx.synchronize()
    assert gpu_ctx.uncertain_count() == 1                      # 40 near-ties around the 10th place: no certificate
    # a wider guard band (tuning key guard_band: 10 + 54 = all 64 list slots) holds the whole cluster: proved exact
    gpu_ctx.set_tuning("guard_band", 54)
    try:
        exp_rows, exp_dist = _oracle_topk(emb, q, 10)
        c.search_topk_device(qd.data_ptr(), 1, 10, 0, o_rows.data_ptr(), o_dist.data_ptr())
        gpu_ctx.synchronize()
        assert gpu_ctx.uncertain_count() == 0
        assert o_rows.cpu().tolist() == exp_rows and np.array_equal(o_dist.cpu().numpy(), exp_dist)
        qs = np.stack([q] * 9)                                 # ... through K3 as well
        got = c.search(qs, top_k=10)
        assert all(g[0].tolist() == exp_rows for g in got)
    finally:
        gpu_ctx.set_tuning("guard_band", 8)
    easy = torch.from_numpy(synth.unit_query(3)[0]).cuda()
    torch.cuda.synchronize()
    c.search_topk_device(easy.data_ptr(), 1, 10, 0, o_rows.data_ptr(), o_dist.data_ptr())
    gpu_ctx.synchronize()
    assert gpu_ctx.uncertain_count() == 0
    c.close()


@pytest.mark.parametrize("nq,at", [(1, 0), (3, 1), (12, 7), (140, 77)])
def test_device_entry_point_says_which_query_it_could_not_prove(gpu_ctx, nq, at):
    """smt_search_topk_device_ex: a verdict PER QUERY (0 proved / 1 certificate failed / 2 overflow) next to the lists, in stream
    order -- search_documents returns a definite list (src/search/mod.rs:107-119), so a pipelined caller must be able to tell WHICH
    answer of WHICH call is not the proved exact top-k.  One adversarial query at index `at` of an otherwise easy batch, through the
    scan kernel (1, 3 queries) and the batched kernel (12: f32-grade nomination, 140: f16 x 2)."""
    import torch
    import semtools_amd as smt

    q, emb, _ = adversarial_corpus(seed=55)
    qs = synth.unit_query(90 + nq, nq=nq)
    qs[at] = q
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    gpu_ctx.uncertain_count()
    qd = torch.from_numpy(qs).cuda()
    o_rows = torch.empty((nq, 10), dtype=torch.int64, device="cuda")
    o_dist = torch.empty((nq, 10), dtype=torch.float64, device="cuda")
    for pinned in (False, True):                               # the verdicts may land in HBM or straight in pinned host memory
        st = torch.full((nq,), 7, dtype=torch.int32)
        st = st.pin_memory() if pinned else st.cuda()
        torch.cuda.synchronize()
        c.search_topk_device(qd.data_ptr(), nq, 10, 0, o_rows.data_ptr(), o_dist.data_ptr(), out_status_ptr=st.data_ptr())
        gpu_ctx.synchronize()
        want = [0] * nq
        want[at] = 1                                           # 40 near-ties around the 10th place: no certificate for THAT query
        assert st.cpu().tolist() == want, (nq, at, pinned)
        assert gpu_ctx.uncertain_count() == 1                  # (the context-wide counter still counts it)
        rows, dist = o_rows.cpu().numpy(), o_dist.cpu().numpy()
        for i in range(nq):
            if i != at:                                        # every answer that says "proved" IS the oracle's
                exp_rows, exp_dist = _oracle_topk(emb, qs[i], 10)
                assert rows[i].tolist() == exp_rows and np.array_equal(dist[i], exp_dist), i
    # the plain entry point is the _ex one without a status buffer
    c.search_topk_device(qd.data_ptr(), nq, 10, 0, o_rows.data_ptr(), o_dist.data_ptr())
    gpu_ctx.synchronize()
    assert gpu_ctx.uncertain_count() == 1
    c.close()


@pytest.mark.parametrize("transport", ["peer", "copy"])
def test_sharded_device_entry_point_reports_the_worst_verdict_of_its_shards(gpu_ctx, transport):
    """smt_sharded_search_topk_device_ex: every rank's status words travel with its k-lists (read in place by the peer transport, in
    the same all-gather otherwise); the device that takes the answer gets, per query, the worst status any shard reported."""
    import torch
    import semtools_amd as smt

    q, emb, cluster = adversarial_corpus(n=6000, n_cluster=150, seed=57)      # ~50 near-ties per shard: no shard can prove its list
    emb[:2000][np.isin(np.arange(2000), cluster)] = synth.unit_rows(2000, seed=5, dup_frac=0, zero_frac=0)[np.isin(np.arange(2000), cluster)]
    # (... except shard 0, whose cluster rows were just replaced: the verdict must come from shards 1 and 2)
    qs = synth.unit_query(95, nq=5)
    qs[3] = q
    n = 3
    g = smt.Group.logical(0, n)
    g.set_transport(transport)
    sc = smt.ShardedCorpus(g, rows=emb)
    qd = torch.from_numpy(qs).cuda()
    outs = [torch.zeros((5, 2, 10), dtype=torch.int64, device="cuda") for _ in range(n)]
    sts = [torch.full((5,), 7, dtype=torch.int32, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    for i in range(n):
        g.ctx(i).uncertain_count()
    # device 0 and device 2 want the answer and the verdicts, device 1 only the answer
    sc.search_topk_device([qd.data_ptr()] * n, 5, 10, [o.data_ptr() for o in outs], [sts[0].data_ptr(), 0, sts[2].data_ptr()])
    # ... and pipelined: the next call (one query, no verdicts wanted) must not disturb the first one's
    sc.search_topk_device([qd.data_ptr()] * n, 1, 10, [0, outs[1].data_ptr(), 0])
    g.synchronize()
    assert sts[0].cpu().tolist() == [0, 0, 0, 1, 0] and sts[2].cpu().tolist() == [0, 0, 0, 1, 0]
    assert sts[1].cpu().tolist() == [7] * 5
    assert g.ctx(0).uncertain_count() == 0 and g.ctx(1).uncertain_count() == 1 and g.ctx(2).uncertain_count() == 1
    m = outs[0].cpu().numpy()
    for i in (0, 1, 2, 4):
        exp_rows, exp_dist = _oracle_topk(emb, qs[i], 10)
        assert np.ascontiguousarray(m[i, 0]).view(np.uint64).tolist() == exp_rows
        assert np.array_equal(np.ascontiguousarray(m[i, 1]).view(np.float64), exp_dist)
    with pytest.raises(RuntimeError):                          # verdicts without the answer: refused
        sc.search_topk_device([qd.data_ptr()] * n, 5, 10, [outs[0].data_ptr(), 0, 0], [0, sts[1].data_ptr(), 0])
    g.synchronize()
    sc.close(); g.close()


def test_many_uncertain_queries_are_reanswered_by_one_batched_pass(gpu_ctx):
    """A batch whose queries ALL sit on near-tie clusters: instead of one exhaustive K4 scan per query the host entry
    point runs ONE batched threshold pass (api.cpp batched_fallback) -- same answers as the oracle, and the pass is
    visible to the profiler as exactly one `gemm_thr` launch.  A band wider than a candidate buffer overflows into
    the per-query route."""
    import semtools_amd as smt

    rng = np.random.default_rng(9)
    emb = synth.unit_rows(6000, seed=31, dup_frac=0.0, zero_frac=0.0)
    queries, spots = [], rng.permutation(6000)
    at = 0
    for c in range(10):                                           # ten queries, each with its own 30-row near-tie cluster
        q = _unit(rng.standard_normal(256))
        v = _unit(q + 0.8 * _unit(rng.standard_normal(256)))
        for i in range(30):
            row = v.copy() if i % 2 else (v * np.float32(0.5 + 0.1 * i)).astype(np.float32)
            if i % 4 == 1:
                idx = rng.choice(256, size=5, replace=False)
                row[idx] = np.nextafter(row[idx], np.float32(np.inf), dtype=np.float32)
            emb[spots[at]] = row
            at += 1
        queries.append(q)
    big_q = _unit(rng.standard_normal(256))                       # one query whose cluster (2500 copies) overflows a buffer
    big_v = _unit(big_q + 0.7 * _unit(rng.standard_normal(256)))
    emb = np.concatenate([emb, np.tile(big_v, (2500, 1))]).astype(np.float32)
    queries.append(big_q)
    queries += list(synth.unit_query(71, nq=3))                   # and three ordinary ones
    qs = np.ascontiguousarray(np.stack(queries), dtype=np.float32)
    c = smt.Corpus(gpu_ctx)
    c.append(np.ascontiguousarray(emb))
    gpu_ctx.set_tuning("fallback_batch_min_rows", 0)
    try:
        gpu_ctx.prof_enable(True)
        gpu_ctx.prof_reset()
        got = c.search(qs, top_k=10)
        launches, _ = gpu_ctx.prof_read("gemm_thr")
        gpu_ctx.prof_enable(False)
    finally:
        gpu_ctx.set_tuning("fallback_batch_min_rows", 100000)
    assert launches == 1
    for i in range(len(qs)):
        exp_rows, exp_dist = _oracle_topk(emb, qs[i], 10)
        assert got[i][0].tolist() == exp_rows, i
        assert np.array_equal(got[i][1], exp_dist), i
    c.close()


def test_sharded_search_redoes_uncertain_queries(gpu_ctx):
    import semtools_amd as smt

    q, emb, _ = adversarial_corpus(seed=66)
    qs = np.concatenate([q[None], synth.unit_query(9, nq=2)])
    g = smt.Group.logical(0, 3)
    sc = smt.ShardedCorpus(g, rows=emb)
    got = sc.search(qs, top_k=10)
    for i in range(len(qs)):
        orows, odist = _oracle_topk(emb, qs[i], 10)
        assert got[i][0].tolist() == orows
        assert np.array_equal(got[i][1], odist)
    sc.close(); g.close()
