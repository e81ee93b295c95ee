"""GPU: small answers are DELIVERED by the select kernel (common.h Delivery; tuning key direct_delivery).

The reference's default workload is one query over the lines of a handful of files (src/cmds/search.rs:245-257), and the agent tool
calls it again and again (src/ask/tools.rs:229-258): a few thousand rows, where a search is its fixed costs.  For host-form top-k
searches with a small answer the last block of the select kernel writes the answer into pinned host memory and a completion word
behind it; the host waits on that word -- no D2H copy command, no hipStreamSynchronize.  Same bar as everywhere: rows and f64
distances = the oracle's, and the same bytes as the copy + synchronise path (direct_delivery = 0).  The route is ASSERTED
(smt_debug_deliveries), not assumed.  (A ONE-launch form -- scan and select in one grid -- was built first and measured slower than
two launches with delivery: the blocks' hand-over inside the grid costs what the second launch does; profiles/r06_small_calls.json.)"""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import synth
from tests.compare import assert_topk_tie_aware, reference_distances

pytestmark = pytest.mark.gpu


def _oracle_topk(emb, q, k):
    res = orc.search_documents(emb, [len(emb)], q, n_lines=0, top_k=k, accurate=True)
    return [r["match_line"] for r in res], np.array([r["distance"] for r in res])


def _fused_launches(ctx):
    return ctx.deliveries()


@pytest.fixture()
def counting(gpu_ctx):
    yield gpu_ctx
    gpu_ctx.set_tuning("direct_delivery", 1)


@pytest.mark.parametrize("n_rows", [1, 2, 3, 4, 5, 63, 64, 65, 1000, 1023, 4096, 16385, 65536])
@pytest.mark.parametrize("nq", [1, 2, 3, 4])
def test_small_answers_are_delivered_and_match_the_oracle(counting, n_rows, nq):
    import semtools_amd as smt

    ctx = counting
    emb = synth.unit_rows(n_rows, seed=100 + n_rows, dup_frac=0.02, zero_frac=0.01)
    qs = synth.unit_query(7 + nq, nq=nq)
    if n_rows > 10:
        qs[0] = emb[n_rows // 3]                                  # an exact hit
    c = smt.Corpus(ctx)
    c.append(emb)
    for k in (1, 3, 10, 56):
        before = _fused_launches(ctx)
        got = c.search(qs, top_k=k)
        assert _fused_launches(ctx) == before + 1, "the answer was not delivered by the select kernel"
        ctx.set_tuning("direct_delivery", 0)
        ref = c.search(qs, top_k=k)
        ctx.set_tuning("direct_delivery", 1)
        assert _fused_launches(ctx) == before + 1
        for i in range(nq):
            rows, dist = _oracle_topk(emb, qs[i], k)
            assert got[i][0].tolist() == rows, (n_rows, nq, k, i)
            assert np.array_equal(got[i][1], dist)
            assert got[i][0].tolist() == ref[i][0].tolist() and np.array_equal(got[i][1], ref[i][1])
        if k == 10 and n_rows >= 1000:   # the tie-aware contract against the serial-f32 restatement
            assert_topk_tie_aware(got[0][0], got[0][1], reference_distances(emb, qs[0]), k)
    c.close()


def test_c1_one_query_over_a_thousand_lines_embedded_on_the_device(counting):
    """BASELINE config c1's shape end to end on the device: 1000 ragged lines pooled by K1 into a corpus, one query pooled the same
    way, top-3 -- one launch, the oracle's rows and distances."""
    import semtools_amd as smt

    ctx = counting
    table = synth.table(3000, seed=5)
    ids, offsets = synth.token_lines(1000, V=3000, seed=6, min_tok=0, max_tok=20)
    model = smt.Model(ctx, table, normalize=True)
    c = smt.Corpus(ctx)
    emb, _ = model.embed(ids, offsets, max_tokens=2048, append_to=c)
    qids = np.array([5, 17, 300, 2999], dtype=np.uint32)
    q, _ = model.embed(qids, np.array([0, 4], dtype=np.uint64), max_tokens=512)
    before = _fused_launches(ctx)
    rows, dist = c.search(q[0], top_k=3)[0]
    assert _fused_launches(ctx) == before + 1
    orows, odist = _oracle_topk(orc.embed_lines(table, ids, offsets, True, 2048), orc.embed_lines(table, qids, np.array([0, 4], dtype=np.uint64), True, 512)[0], 3)
    assert rows.tolist() == orows and np.array_equal(dist, odist)
    c.close(); model.close()


def test_workspace_threshold_and_zero_query_with_delivery(counting):
    import semtools_amd as smt
    from semtools_amd import _lib as L

    ctx = counting
    emb = synth.unit_rows(5000, seed=31, dup_frac=0.05, zero_frac=0.02)
    c = smt.Corpus(ctx)
    c.append(emb)
    qs = synth.unit_query(12, nq=3)
    qs[1] = 0.0                                                   # zero query: zero rows first (distance 0), everything else 1
    row_path = np.zeros(len(emb), dtype=np.uint32)
    row_line = np.arange(len(emb), dtype=np.int32)
    for thr in (0.5, 0.9, 1.5):
        before = _fused_launches(ctx)
        got = c.search(qs, top_k=8, mode=L.MODE_WORKSPACE, max_distance=thr)
        assert _fused_launches(ctx) == before + 1
        for i in range(3):
            ref = orc.search_line_embeddings(emb, row_path, row_line, qs[i], [0], 8, thr)
            assert got[i][0].tolist() == [r["row"] for r in ref], (thr, i)
            assert np.allclose(got[i][1].astype(np.float32), [r["distance"] for r in ref], rtol=0, atol=1e-5)
    got = c.search(qs, top_k=5)
    for i in range(3):
        rows, dist = _oracle_topk(emb, qs[i], 5)
        assert got[i][0].tolist() == rows and np.array_equal(got[i][1], dist)
    # workspace mode without a threshold and over a path subset; the ZERO query's answer is qdrant's (every point scores 0: the first
    # rows of the subset at distance 1.0), not search_documents' (zero rows first at distance 0) -- on one GPU and on three shards
    ranges = [(10, 60), (2000, 2004), (4000, 5000)]
    row_path[:] = 0
    for a, b in ranges:
        row_path[a:b] = 1
    g = smt.Group.logical(0, 3)
    sc = smt.ShardedCorpus(g, rows=emb)
    for searcher in (c, sc):
        got = searcher.search(qs, top_k=6, mode=L.MODE_WORKSPACE, ranges=ranges)
        for i in range(3):
            ref = orc.search_line_embeddings(emb, row_path, row_line, qs[i], [1], 6, None)
            assert got[i][0].tolist() == [r["row"] for r in ref], i
            assert np.allclose(got[i][1].astype(np.float32), [r["distance"] for r in ref], rtol=0, atol=1e-5)
        assert got[1][0].tolist() == [10, 11, 12, 13, 14, 15] and got[1][1].tolist() == [1.0] * 6
    sc.close(); g.close()
    c.close()


def test_near_ties_beyond_the_guard_band_are_re_answered_exhaustively(counting):
    """40 rows within 1e-7 of each other around the k-th place: the delivered answer says "uncertain", the call falls back to the
    exhaustive re-answer and still returns the oracle's list."""
    import semtools_amd as smt
    from tests.test_gpu_nearties import adversarial_corpus

    ctx = counting
    q, emb, _ = adversarial_corpus(seed=55)
    c = smt.Corpus(ctx)
    c.append(emb)
    before = _fused_launches(ctx)
    rows, dist = c.search(q, top_k=10)[0]
    assert _fused_launches(ctx) == before + 1
    orows, odist = _oracle_topk(emb, q, 10)
    assert rows.tolist() == orows and np.array_equal(dist, odist)
    c.close()


def test_what_qualifies_and_what_keeps_the_copy(counting):
    import semtools_amd as smt

    ctx = counting
    emb = synth.unit_rows(70000, seed=41)
    c = smt.Corpus(ctx)
    c.append(emb)
    qs = synth.unit_query(3, nq=40)
    before = _fused_launches(ctx)
    c.search(qs[0], top_k=57)                                     # large k: all keys + radix sort, host-side ordering
    c.search(qs[0], max_distance=0.8)                             # threshold mode (every hit): unbounded answers
    c.search(qs, top_k=3)                                         # 40 queries: beyond one select launch's worth of delivery
    c.search(qs[:20], top_k=56)                                   # 20 x 56 x 16 B: more than 8 KiB of answer
    assert _fused_launches(ctx) == before
    for kw in (dict(top_k=3), dict(top_k=3, ranges=[(0, 1000), (5000, 70000)]), dict(top_k=56)):
        got = c.search(qs[:5], **kw)                              # the scan kernel, the batched kernel, range-filtered: all deliver
        keep = np.arange(len(emb)) if "ranges" not in kw else np.concatenate([np.arange(a, b) for a, b in kw["ranges"]])
        for i in range(5):
            orows, odist = _oracle_topk(emb[keep], qs[i], kw["top_k"])
            assert got[i][0].tolist() == keep[orows].tolist() and np.array_equal(got[i][1], odist)
    assert _fused_launches(ctx) == before + 3
    c.close()


def test_a_thousand_small_calls_in_a_row_with_other_calls_in_between(counting):
    """The completion word is per call (a sequence number): interleaving fused calls with device-form searches, appends and calls on
    a second corpus of the same context never lets one call read another's answer."""
    import semtools_amd as smt

    ctx = counting
    a_rows = synth.unit_rows(3000, seed=51)
    b_rows = synth.unit_rows(700, seed=52)
    a, b = smt.Corpus(ctx), smt.Corpus(ctx)
    a.append(a_rows[:2000]); b.append(b_rows)
    qs = synth.unit_query(9, nq=40)
    want_a = {}
    for i in range(1000):
        q = qs[i % 40]
        if i == 500:
            a.append(a_rows[2000:])
            want_a.clear()
        n = a.rows
        if (i % 40, n) not in want_a:
            want_a[(i % 40, n)] = _oracle_topk(a_rows[:n], q, 4)
        rows, dist = a.search(q, top_k=4)[0]
        assert rows.tolist() == want_a[(i % 40, n)][0] and np.array_equal(dist, want_a[(i % 40, n)][1]), i
        if i % 7 == 0:
            rows, dist = b.search(qs[(i + 1) % 40], top_k=2)[0]
            orows, odist = _oracle_topk(b_rows, qs[(i + 1) % 40], 2)
            assert rows.tolist() == orows and np.array_equal(dist, odist)
    a.close(); b.close()


def test_the_index_search_delivers_small_answers_too(counting):
    """smt_ivfpq_search (host form) takes the same way home: the select stage of the index search writes a small answer into pinned
    memory and the host waits on its completion word; same bytes as the copy + synchronise path, large answers keep the copy."""
    import semtools_amd as smt
    from tests.test_gpu_ivfpq import clustered

    ctx = counting
    x, _ = clustered(30000, 128, seed=4)
    q, _ = clustered(64, 128, seed=5)
    c = smt.Corpus(ctx)
    c.append(x)
    ix = smt.IvfPq(c, nlist=64, train_iters=5, local_pca=True)
    for nq, k in ((1, 10), (4, 3), (32, 10), (64, 10), (20, 56)):
        before = _fused_launches(ctx)
        got = ix.search(q[:nq], top_k=k, nprobe=8, rerank=128)
        delivered = _fused_launches(ctx) - before
        assert delivered == (1 if nq <= 32 and nq * k * 16 + nq * 8 <= 8192 else 0), (nq, k, delivered)
        ctx.set_tuning("direct_delivery", 0)
        ref = ix.search(q[:nq], top_k=k, nprobe=8, rerank=128)
        ctx.set_tuning("direct_delivery", 1)
        for (r1, d1), (r2, d2) in zip(got, ref):
            assert r1.tolist() == r2.tolist() and np.array_equal(d1, d2)
        for i in range(min(nq, 4)):   # every returned pair is exact
            assert np.array_equal(got[i][1], np.array([orc.cosine(q[i], x[int(r)], accurate=True) for r in got[i][0]]))
    ix.close(); c.close()
