"""GPU parity: K2 scan + top-k / threshold selection vs the CPU oracle (A5 + A6 + A10).

Bar: indices bit-exact, distances within 1e-5.  The
library does better -- every returned distance is an f64 re-evaluation in index
order, so we also assert (near) bit-equality with the oracle's "accurate" mode.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import synth
from tests.compare import assert_topk_tie_aware, reference_distances

pytestmark = pytest.mark.gpu


def _oracle_topk(emb, q, k, accurate=True):
    res = orc.search_documents(emb, [len(emb)], q, n_lines=0, top_k=k, accurate=accurate)
    return np.array([r["match_line"] for r in res], dtype=np.uint64), np.array([r["distance"] for r in res])


@pytest.fixture(scope="module")
def corpus20k(gpu_ctx):
    import semtools_amd as smt

    emb = synth.unit_rows(20000, seed=3)
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    yield emb, c
    c.close()


@pytest.mark.parametrize("k", [1, 3, 10, 56, 64])
def test_topk_matches_oracle(corpus20k, k):
    emb, c = corpus20k
    q = synth.unit_query(4)[0]
    rows, dist = c.search(q, top_k=k)[0]
    orows, odist = _oracle_topk(emb, q, k)
    assert rows.tolist() == orows.tolist()
    np.testing.assert_allclose(dist, odist, rtol=0, atol=1e-12)
    # the contract proper: against the serial-f32 restatement -- the closest thing to what the
    # reference executes -- indices are exact up to permutations inside < 1e-5 tie groups, distances within 1e-5
    srows, sdist = _oracle_topk(emb, q, k, accurate=False)
    np.testing.assert_allclose(dist, sdist, rtol=0, atol=1e-5)
    permuted = assert_topk_tie_aware(rows, dist, reference_distances(emb, q, accurate=False), k)
    assert permuted == 0 or srows.tolist() != rows.tolist()   # on this corpus the two orders normally coincide


def test_distances_bit_exact_vs_accurate_oracle(corpus20k):
    emb, c = corpus20k
    q = synth.unit_query(11)[0]
    rows, dist = c.search(q, top_k=64)[0]
    ref = np.array([orc.cosine(q, emb[int(r)], accurate=True) for r in rows])
    assert np.array_equal(dist, ref), f"max |diff| = {np.abs(dist - ref).max()}"


def test_duplicates_and_zero_rows_order(gpu_ctx):
    """Exact duplicate rows tie bit-for-bit and must come back in row order (stable sort,
    src/search/mod.rs:107-111); zero rows have distance exactly 1 (simsimd ab==0 rule)."""
    import semtools_amd as smt

    rng = np.random.default_rng(7)
    base = synth.unit_rows(3000, seed=8, dup_frac=0, zero_frac=0)
    q = synth.unit_query(9)[0]
    emb = base.copy()
    emb[[5, 77, 300, 2999]] = q                     # 4 copies of the query itself -> distance 0
    emb[[10, 20]] = 0.0
    near = (q + 0.01 * rng.standard_normal(256)).astype(np.float32)
    emb[[1500, 40, 2200]] = near                    # triple tie, must be returned as 40, 1500, 2200
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    rows, dist = c.search(q, top_k=8)[0]
    orows, odist = _oracle_topk(emb, q, 8)
    assert rows.tolist() == orows.tolist()
    assert rows[:7].tolist() == [5, 77, 300, 2999, 40, 1500, 2200]
    assert np.array_equal(dist, odist)
    # zero rows: query them explicitly through a range filter
    r2, d2 = c.search(q, top_k=3, ranges=[(10, 11), (20, 21)])[0]
    assert r2.tolist() == [10, 20] and d2.tolist() == [1.0, 1.0]
    # zero query: zero rows are distance 0, everything else 1 (a2==0 && b2==0 -> 0 ; ab==0 -> 1)
    z = np.zeros(256, np.float32)
    r3, d3 = c.search(z, top_k=4)[0]
    o3, od3 = _oracle_topk(emb, z, 4)
    assert r3.tolist() == o3.tolist() == [10, 20, 0, 1]
    assert d3.tolist() == od3.tolist() == [0.0, 0.0, 1.0, 1.0]
    c.close()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 511, 513, 4097])
def test_ragged_sizes(gpu_ctx, n):
    import semtools_amd as smt

    emb = synth.unit_rows(n, seed=100 + n, dup_frac=0.05, zero_frac=0.0)
    q = synth.unit_query(5)[0]
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    for k in (1, 3, 10):
        rows, dist = c.search(q, top_k=k)[0]
        orows, odist = _oracle_topk(emb, q, k)
        assert rows.tolist() == orows.tolist(), (n, k)
        assert np.array_equal(dist, odist)
    c.close()


def test_empty_corpus_and_topk_zero(gpu_ctx):
    import semtools_amd as smt

    c = smt.Corpus(gpu_ctx)
    q = synth.unit_query(5)[0]
    assert c.search(q, top_k=3)[0][0].size == 0                       # empty docs -> empty (mod.rs:390)
    c.append(synth.unit_rows(10, seed=1))
    assert c.search(q, top_k=0)[0][0].size == 0                       # take(0)
    assert c.search(q, top_k=0, max_distance=0.5, mode=1)[0][0].size == 0  # store.rs:489-491
    c.close()


def test_threshold_mode_returns_all_under_max_distance(corpus20k):
    """mod.rs:88-89,115-116: strict <, every hit returned, top_k ignored."""
    emb, c = corpus20k
    q = synth.unit_query(4)[0]
    for thr in (0.85, 0.9, 1.0):
        rows, dist = c.search(q, top_k=3, max_distance=thr)[0]
        res = orc.search_documents(emb, [len(emb)], q, n_lines=0, top_k=3, max_distance=thr, accurate=True)
        orows = [r["match_line"] for r in res]
        assert rows.tolist() == orows, thr
        assert np.array_equal(dist, np.array([r["distance"] for r in res]))
        assert (dist < thr).all()


def test_threshold_mode_several_queries_take_one_batched_sweep(corpus20k):
    """Several threshold queries on one unfiltered shard are answered by ONE sweep of the batched kernel (api.cpp,
    batched_fallback with strict = true); a query with more hits than a candidate buffer (2048 rows) still takes the
    streaming K4 scan.  Same hit lists as the oracle for both kinds."""
    emb, c = corpus20k
    qs = synth.unit_query(31, nq=6)
    c.ctx.set_tuning("fallback_batch_min_rows", 0)
    try:
        for thr in (0.82, 0.88, 0.97):                       # ~10, ~300, ~6000 hits per query: the last one overflows
            c.ctx.prof_enable(True)
            c.ctx.prof_reset()
            got = c.search(qs, top_k=3, max_distance=thr)
            launches, _ = c.ctx.prof_read("gemm_thr")
            c.ctx.prof_enable(False)
            assert launches == 1
            for i in range(len(qs)):
                res = orc.search_documents(emb, [len(emb)], qs[i], n_lines=0, top_k=3, max_distance=thr, accurate=True)
                assert got[i][0].tolist() == [r["match_line"] for r in res], (thr, i)
                assert np.array_equal(got[i][1], np.array([r["distance"] for r in res]))
    finally:
        c.ctx.set_tuning("fallback_batch_min_rows", 100000)


def test_threshold_truncation_reports_true_count(corpus20k):
    import ctypes as C
    from semtools_amd import _lib as L

    emb, c = corpus20k
    q = synth.unit_query(4)
    out_rows = np.zeros((1, 5), np.uint64)
    out_dist = np.zeros((1, 5), np.float64)
    counts = np.zeros(1, np.uint64)
    rc = L.lib().smt_search(c._h, L.np_ptr(q), 1, 3, 0.9, 0, None, 0, 0, L.np_ptr(out_rows), L.np_ptr(out_dist),
                            L.np_ptr(counts), 5)
    full = c.search(q[0], max_distance=0.9)[0]
    assert rc == L.SMT_E_TRUNCATED
    assert int(counts[0]) == full[0].size > 5
    assert out_rows[0].tolist() == full[0][:5].tolist()


def test_multi_query_small_batches(corpus20k):
    emb, c = corpus20k
    qs = synth.unit_query(21, nq=7)
    got = c.search(qs, top_k=5)
    for i in range(7):
        orows, odist = _oracle_topk(emb, qs[i], 5)
        assert got[i][0].tolist() == orows.tolist()
        assert np.array_equal(got[i][1], odist)


def test_range_filter_and_row_base(corpus20k):
    emb, c = corpus20k
    q = synth.unit_query(4)[0]
    ranges = [(100, 164), (1000, 1001), (5000, 9000), (19990, 20000)]
    rows, dist = c.search(q, top_k=10, ranges=ranges, row_base=1_000_000_000_000)[0]
    mask = np.zeros(len(emb), bool)
    for b, e in ranges:
        mask[b:e] = True
    idx = np.nonzero(mask)[0]
    orows, odist = _oracle_topk(emb[idx], q, 10)
    assert (rows - 1_000_000_000_000).tolist() == idx[orows.astype(np.int64)].tolist()
    assert np.array_equal(dist, odist)


def test_workspace_mode_matches_store_semantics(corpus20k):
    """A10 (store.rs:481-546): score > 1 - max_distance, then ALWAYS top_k."""
    emb, c = corpus20k
    q = synth.unit_query(4)[0]
    row_path = (np.arange(len(emb)) // 100).astype(np.uint32)      # 200 docs x 100 lines
    row_line = (np.arange(len(emb)) % 100).astype(np.int32)
    subset = np.array([3, 17, 18, 150], np.uint32)
    ranges = [(int(p) * 100, int(p) * 100 + 100) for p in subset]
    for thr in (None, 0.95, 0.5):
        rows, dist = c.search(q, top_k=4, max_distance=thr, mode=1, ranges=ranges)[0]
        ref = orc.search_line_embeddings(emb, row_path, row_line, q, subset, 4, thr)
        assert rows.tolist() == [r["row"] for r in ref], thr
        np.testing.assert_allclose(dist.astype(np.float32), [r["distance"] for r in ref], atol=1e-5)


def test_reference_store_known_answer(gpu_ctx):
    """src/workspace/store.rs:814-850: stored [0.1;256], [0.5;256], [0.75;256]; query [0.1;256],
    subset = doc1 only, k=1, max_distance 0.1 -> exactly (doc1, line 0, distance < 0.1)."""
    import semtools_amd as smt

    emb = np.stack([np.full(256, v, np.float32) for v in (0.1, 0.5, 0.75)])
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    rows, dist = c.search(np.full(256, 0.1, np.float32), top_k=1, max_distance=0.1, mode=1, ranges=[(0, 1)])[0]
    assert rows.tolist() == [0] and dist[0] < 0.1
    c.close()


def test_deterministic(corpus20k):
    emb, c = corpus20k
    q = synth.unit_query(4)[0]
    a = c.search(q, top_k=10)[0]
    b = c.search(q, top_k=10)[0]
    assert a[0].tolist() == b[0].tolist() and np.array_equal(a[1], b[1])


def test_shard_merge_equals_single(gpu_ctx, corpus20k):
    """Row-sharding + merge of per-shard top-k == single-shard result."""
    import semtools_amd as smt

    emb, c = corpus20k
    q = synth.unit_query(4)[0]
    want = c.search(q, top_k=10)[0]
    bounds = [0, 3000, 3001, 12000, 20000]
    rows, dist = [], []
    for b, e in zip(bounds[:-1], bounds[1:]):
        s = smt.Corpus(gpu_ctx)
        s.append(emb[b:e])
        r, d = s.search(q, top_k=10, row_base=b)[0]
        pr = np.full(10, np.iinfo(np.uint64).max, np.uint64)
        pd = np.full(10, np.inf)
        pr[: r.size], pd[: d.size] = r, d
        rows.append(pr[None])
        dist.append(pd[None])
        s.close()
    mr, md, cnt = smt.merge_topk(np.stack(rows), np.stack(dist), 10)
    assert mr[0].tolist() == want[0].tolist() and np.array_equal(md[0], want[1])


@pytest.mark.parametrize("k", [65, 200, 1000])
def test_large_top_k_falls_back_to_sort_path(corpus20k, k):
    """top_k > 64 (the reference takes any k after sorting everything, mod.rs:107-119)."""
    emb, c = corpus20k
    q = synth.unit_query(4)[0]
    rows, dist = c.search(q, top_k=k)[0]
    orows, odist = _oracle_topk(emb, q, k)
    assert rows.tolist() == orows.tolist()
    assert np.array_equal(dist, odist)


def test_large_top_k_with_ranges_and_more_than_rows(corpus20k):
    emb, c = corpus20k
    q = synth.unit_query(4)[0]
    rows, dist = c.search(q, top_k=500, ranges=[(100, 300), (1000, 1100)])[0]
    idx = np.r_[100:300, 1000:1100]
    orows, odist = _oracle_topk(emb[idx], q, 500)
    assert rows.size == 300 and rows.tolist() == idx[orows.astype(np.int64)].tolist()
    assert np.array_equal(dist, odist)


def test_device_merge_kernels_match_host_merge(gpu_ctx, corpus20k):
    """merge_topk_kernel (plain and packed layouts) == the host merge == single-shard search."""
    import torch
    import semtools_amd as smt

    emb, c = corpus20k
    qs = synth.unit_query(31, nq=3)
    k = 6
    bounds = [0, 7000, 7001, 20000]
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream()
    ctx = smt.Context(0, stream=stream.cuda_stream)
    qd = torch.from_numpy(qs).to(dev)
    packed = torch.empty((len(bounds) - 1, 3, 2, k), dtype=torch.int64, device=dev)
    shards = []
    for li, (b, e) in enumerate(zip(bounds[:-1], bounds[1:])):
        x = torch.from_numpy(emb[b:e]).to(dev)
        shards.append(x)
        s = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=e - b)
        loc = torch.empty((3, 2, k), dtype=torch.int64, device=dev)
        for qi in range(3):  # K2 path per query, rows/dist of a query share one packed slot
            s.search_topk_device(qd[qi].data_ptr(), 1, k, b, loc[qi, 0].data_ptr(), loc[qi, 1].data_ptr())
        packed[li].copy_(loc)
        s.close()
    outp = torch.empty((3, 2, k), dtype=torch.int64, device=dev)
    ctx.merge_topk_packed_device(packed.data_ptr(), len(bounds) - 1, 3, k, k, outp.data_ptr())
    rows_plain = packed[:, :, 0, :].contiguous()
    dist_plain = packed[:, :, 1, :].contiguous()
    o_rows = torch.empty((3, k), dtype=torch.int64, device=dev)
    o_dist = torch.empty((3, k), dtype=torch.float64, device=dev)
    ctx.merge_topk_device(rows_plain.data_ptr(), dist_plain.data_ptr(), len(bounds) - 1, 3, k, k, o_rows.data_ptr(), o_dist.data_ptr())
    torch.cuda.synchronize()
    hr, hd, _ = smt.merge_topk(rows_plain.cpu().numpy().view(np.uint64), dist_plain.view(torch.float64).cpu().numpy(), k)
    want = c.search(qs, top_k=k)
    for qi in range(3):
        assert outp[qi, 0].cpu().tolist() == o_rows[qi].cpu().tolist() == hr[qi].astype(np.int64).tolist() == want[qi][0].astype(np.int64).tolist()
        assert np.array_equal(outp[qi, 1].cpu().numpy().view(np.float64), want[qi][1])
        assert np.array_equal(o_dist[qi].cpu().numpy(), hd[qi])
    ctx.close()


def test_corpus_save_load_and_incremental_append(gpu_ctx, tmp_path):
    """smt_corpus_save / _load round trip and the O(new rows) flush used by the workspace store."""
    import semtools_amd as smt

    a = synth.unit_rows(1000, seed=1)
    b = synth.unit_rows(337, seed=2)
    c = smt.Corpus(gpu_ctx)
    c.append(a)
    f = tmp_path / "rows.f32"
    c.save(f)
    c.append(b)
    c.append_to_file(f, 1000)
    d = smt.Corpus.load(gpu_ctx, f)
    assert d.rows == 1337 and np.array_equal(d.read_rows(0, 1337), np.concatenate([a, b]))
    with pytest.raises(smt.SmtError):
        c.append_to_file(f, 1000)            # the file no longer holds exactly 1000 rows
    q = synth.unit_query(3)[0]
    assert d.search(q, top_k=5)[0][0].tolist() == c.search(q, top_k=5)[0][0].tolist()
    c.close(); d.close()


def test_error_codes_and_argument_validation(gpu_ctx, tmp_path):
    """Error behaviour of the ABI: status codes + message, nothing aborts."""
    import ctypes as C
    import semtools_amd as smt
    from semtools_amd import _lib as L

    lib = L.lib()
    h = C.c_void_p()
    assert lib.smt_corpus_create(gpu_ctx._h, 128, 0, C.byref(h)) == L.SMT_E_UNSUPPORTED       # dim is fixed at 256
    assert b"256" in lib.smt_last_error()
    assert lib.smt_corpus_load(gpu_ctx._h, str(tmp_path / "missing.f32").encode(), C.byref(h)) == L.SMT_E_IO
    (tmp_path / "junk.f32").write_bytes(b"not a corpus file at all, just bytes" * 4)
    assert lib.smt_corpus_load(gpu_ctx._h, str(tmp_path / "junk.f32").encode(), C.byref(h)) == L.SMT_E_IO
    c = smt.Corpus(gpu_ctx)
    c.append(synth.unit_rows(100, seed=1))
    q = synth.unit_query(1)[0]
    for bad in ([(10, 5)], [(0, 101)], [(50, 60), (55, 70)], [(60, 70), (10, 20)]):
        with pytest.raises(smt.SmtError) as e:
            c.search(q, top_k=3, ranges=bad)
        assert e.value.code == L.SMT_E_INVALID
    with pytest.raises(smt.SmtError):
        c.read_rows(90, 20)
    with pytest.raises(smt.SmtError):
        c.truncate(101)
    c.truncate(40)
    assert c.rows == 40 and c.search(q, top_k=100)[0][0].size == 40
    assert c.search(q, top_k=3, ranges=[(5, 5)])[0][0].size == 0                                # empty range -> nothing
    table = synth.table(10, seed=1)
    m = smt.Model(gpu_ctx, table)
    with pytest.raises(smt.SmtError):
        m.embed(np.array([1, 2, 3], np.uint32), np.array([0, 3, 2], np.uint64))                 # decreasing offsets
    out, _ = m.embed(np.array([1, 999, 2], np.uint32), np.array([0, 3], np.uint64))             # id >= V contributes nothing
    assert np.array_equal(out, orc.embed_lines(table, np.array([1, 999, 2], np.uint32), np.array([0, 3], np.uint64)))
    m.close(); c.close()


def test_many_small_ranges_all_paths(corpus20k):
    """Path-subset search (store.rs:507-515): hundreds of ranges whose lengths are not multiples of the
    4-row chunk the filtered kernels stream (1, 2, 3, 5, 7 ... rows, adjacent and far apart), through the
    single-query scan, the 2- and 4-query scans, the threshold scan and the large-k path; the answer must
    equal the oracle run on the gathered subset."""
    emb, c = corpus20k
    rng = np.random.default_rng(21)
    ranges, pos = [], 0
    while pos < 19_000:
        length = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 13, 64, 100]))
        ranges.append((pos, pos + length))
        pos += length + int(rng.choice([0, 1, 3, 50, 400]))       # gap 0 = adjacent ranges
    assert len(ranges) > 100
    idx = np.concatenate([np.arange(b, e) for b, e in ranges])
    sub = emb[idx]
    qs = synth.unit_query(4, nq=7)
    for nq in (1, 2, 3, 4, 7):                                       # 7 = 4 + 2 + 1 query launches
        got = c.search(qs[:nq], top_k=10, ranges=ranges)
        for qi in range(nq):
            orows, odist = _oracle_topk(sub, qs[qi], 10)
            assert got[qi][0].tolist() == idx[orows.astype(np.int64)].tolist(), (nq, qi)
            assert np.array_equal(got[qi][1], odist)
    # threshold mode: few hits (host-ordered) and many hits (device-ordered, > 2048)
    for md in (0.85, 1.06):
        rows, dist = c.search(qs[0], max_distance=md, ranges=ranges)[0]
        res = orc.search_documents(sub, [len(sub)], qs[0], n_lines=0, top_k=3, max_distance=md, accurate=True)
        assert rows.tolist() == [int(idx[r["match_line"]]) for r in res], md
        assert np.array_equal(dist, np.array([r["distance"] for r in res]))
    assert len(c.search(qs[0], max_distance=1.06, ranges=ranges)[0][0]) > 2048
    # large k (> 64) keeps its own per-row range lookup
    rows, dist = c.search(qs[0], top_k=200, ranges=ranges)[0]
    orows, odist = _oracle_topk(sub, qs[0], 200)
    assert rows.tolist() == idx[orows.astype(np.int64)].tolist() and np.array_equal(dist, odist)


def test_threshold_many_hits_device_order_matches_oracle(corpus20k):
    """More than 2048 hits take the device-side ordering (radix sort by row, exact rescoring, stable radix sort
    by distance); exact duplicates must come back in row order."""
    emb, c = corpus20k
    x = emb.copy()
    x[5000:5600] = x[100:700]                                        # 600 exact duplicates -> distance ties
    import semtools_amd as smt
    c2 = smt.Corpus(c.ctx)
    c2.append(x)
    q = synth.unit_query(4)[0]
    for md in (0.95, 1.02):
        rows, dist = c2.search(q, max_distance=md)[0]
        res = orc.search_documents(x, [len(x)], q, n_lines=0, top_k=3, max_distance=md, accurate=True)
        assert len(res) > 2048
        assert rows.tolist() == [r["match_line"] for r in res], md
        assert np.array_equal(dist, np.array([r["distance"] for r in res]))
    c2.close()


def test_async_select_pipeline_equals_stream_order(gpu_ctx):
    """async_select: the select of query i runs on the aux stream while query i+1 scans (device-scope flags).
    Same answers as the in-order path, for a long back-to-back series, across a change of k and corpus, and
    after switching the mode off again (drain)."""
    import torch
    import semtools_amd as smt

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    x = torch.randn(300_000, 256, device=dev, generator=g)
    x /= x.norm(dim=1, keepdim=True)
    qs = torch.randn(64, 256, device=dev, generator=g)
    torch.cuda.synchronize()
    ctx = smt.Context(0)
    big = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=300_000)
    small = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=5_000)

    def series(corpus, k, n):
        rows = torch.full((n, k), -7, dtype=torch.int64, device=dev)
        dist = torch.full((n, k), -7.0, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        for i in range(n):
            corpus.search_topk_device(qs[i % 64].data_ptr(), 1, k, 0, rows[i].data_ptr(), dist[i].data_ptr())
        ctx.synchronize()
        return rows.cpu().numpy(), dist.cpu().numpy()

    want = {(name, k): series(c, k, 200) for name, c in (("big", big), ("small", small)) for k in (10, 3)}
    ctx.set_tuning("async_select", 1)
    for rep in range(2):
        for (name, k), (wr, wd) in want.items():
            gr, gd = series(big if name == "big" else small, k, 200)
            assert np.array_equal(gr, wr) and np.array_equal(gd, wd), (rep, name, k)
    # a host-level search in between must see finished results and must not be async itself
    r, d = big.search(qs[0].cpu().numpy(), top_k=10)[0]
    assert r.tolist() == want[("big", 10)][0][0].tolist()
    ctx.set_tuning("async_select", 0)
    gr, gd = series(big, 10, 50)
    assert np.array_equal(gr, want[("big", 10)][0][:50])
    big.close(); small.close(); ctx.close()


@pytest.mark.parametrize("steal,pct", [(1, 3), (2, 6), (4, 6), (4, 50), (8, 10), (16, 25)])
def test_rows_dealt_while_the_kernel_runs_change_no_answer(gpu_ctx, steal, pct):
    """K2's dynamic deal (tuning keys scan_steal / scan_steal_pct; off by default): which block reduces a row cannot change the answer
    -- the union of the blocks' k' best holds the k' best rows whatever the deal.  1, 2, 3 and 4 queries over enough rows that the
    dynamic groups are in play, against the static deal (bytes) and the oracle."""
    import semtools_amd as smt

    n = 600_000
    emb = synth.unit_rows(n, seed=17, dup_frac=0.001, zero_frac=0.0005)
    qs = synth.unit_query(23, nq=4)
    qs[2] = emb[n - 7]                                            # an exact hit in the very last (dynamic) group
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    gpu_ctx.set_tuning("gemm_min_nq", 8)                         # (keep 3 and 4 queries on the scan kernel)
    try:
        want = {nq: c.search(qs[:nq], top_k=10) for nq in (1, 2, 3, 4)}
        gpu_ctx.set_tuning("scan_steal", steal)
        gpu_ctx.set_tuning("scan_steal_pct", pct)
        for rep in range(3):
            for nq in (1, 2, 3, 4):
                got = c.search(qs[:nq], top_k=10)
                for i in range(nq):
                    assert got[i][0].tolist() == want[nq][i][0].tolist() and np.array_equal(got[i][1], want[nq][i][1]), (rep, nq, i)
    finally:
        gpu_ctx.set_tuning("scan_steal", 0)
        gpu_ctx.set_tuning("scan_steal_pct", 6)
        gpu_ctx.set_tuning("gemm_min_nq", 5)
    res = orc.search_documents(emb, [n], qs[2], n_lines=0, top_k=10, accurate=True)
    assert want[3][2][0].tolist() == [r["match_line"] for r in res]
    c.close()
