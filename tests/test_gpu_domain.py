"""GPU: the library's numeric DOMAIN (semtools_amd/csrc/domain.hip, "Domain" in include/semtools_hip.h).

The reference has no definite answer for rows or queries with non-finite components (cos_finish turns the NaN into distance 0.0, the
best score: src/search/mod.rs:86-89, 107-111) or with magnitudes that overflow / underflow simsimd's f32 accumulators
(tests/test_oracle.py documents both).  The boundary therefore REFUSES such vectors -- loudly, where they would enter -- and inside
the domain (largest magnitude of a vector 0 or within [2^-40, 2^40]) every kernel family must answer exactly as for unit rows:
indices and f64 distances bit-equal to the oracle's accurate form, and a power-of-two scale of any row or query changes nothing."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import synth

pytestmark = pytest.mark.gpu

BAD_ROWS = {
    "nan": lambda r: _poke(r, 17, np.nan),
    "+inf": lambda r: _poke(r, 200, np.inf),
    "-inf": lambda r: _poke(r, 0, -np.inf),
    "huge": lambda r: _poke(r, 255, 1e20),                       # its square overflows f32
    "above-2^40": lambda r: _poke(r, 3, 2.0 ** 40 * 1.0001),
    "tiny": lambda r: (r * np.float32(1e-25)).astype(np.float32),  # every square underflows f32
    "below-2^-40": lambda r: (r / np.abs(r).max() * np.float32(2.0 ** -40 * 0.999)).astype(np.float32),
    "denormal": lambda r: _only(r, 9, 1e-42),
}


def _poke(r, i, v):
    r = r.copy()
    r[i] = v
    return r


def _only(r, i, v):
    r = np.zeros_like(r)
    r[i] = v
    return r


def _refused(fn, code=-1):
    import semtools_amd as smt  # noqa: F401
    from semtools_amd import _lib as L

    with pytest.raises(L.SmtError) as e:
        fn()
    assert e.value.code == code, e.value
    return str(e.value)


@pytest.mark.parametrize("kind", list(BAD_ROWS))
def test_rows_outside_the_domain_do_not_enter_a_corpus(gpu_ctx, kind, tmp_path):
    import torch
    import semtools_amd as smt

    emb = synth.unit_rows(3000, seed=11)
    bad = emb.copy()
    bad[1234] = BAD_ROWS[kind](emb[1234])
    bad[2999] = BAD_ROWS[kind](emb[2999])
    c = smt.Corpus(gpu_ctx)
    c.append(emb[:500])
    msg = _refused(lambda: c.append(bad))                       # smt_corpus_append_host
    assert "2 of 3000 rows" in msg and "first: row 1234" in msg, msg
    assert c.rows == 500 and np.array_equal(c.read_rows(0, 500), emb[:500])
    msg = _refused(lambda: c.write_rows(100, bad[1230:1240]))   # smt_corpus_write_rows: nothing overwritten
    assert "row 4 of the 10" in msg, msg
    assert np.array_equal(c.read_rows(0, 500), emb[:500])
    t = torch.from_numpy(bad).cuda()
    torch.cuda.synchronize()
    msg = _refused(lambda: smt.Corpus(gpu_ctx, device_ptr=t.data_ptr(), rows=len(bad)))   # smt_corpus_from_device
    assert "first: row 1234" in msg, msg
    # a damaged corpus file: a good one with the bad rows written over its payload
    c2 = smt.Corpus(gpu_ctx)
    c2.append(emb)
    path = tmp_path / "rows.f32"
    c2.save(path)
    raw = bytearray(path.read_bytes())
    raw[32 + 1234 * 1024: 32 + 1235 * 1024] = bad[1234].tobytes()
    path.write_bytes(bytes(raw))
    msg = _refused(lambda: smt.Corpus.load(gpu_ctx, path), code=-5)      # SMT_E_IO, like a truncated file
    assert "row 1234 of the file" in msg, msg
    # ... and the corpus that refused is as good as before
    q = synth.unit_query(3)[0]
    rows, dist = c.search(q, top_k=5)[0]
    ref = orc.search_documents(emb[:500], [500], q, n_lines=0, top_k=5, accurate=True)
    assert rows.tolist() == [r["match_line"] for r in ref] and np.array_equal(dist, [r["distance"] for r in ref])
    c.close(); c2.close()


def test_rows_pooled_from_a_damaged_table_do_not_enter_a_corpus(gpu_ctx):
    import semtools_amd as smt

    table = synth.table(500, seed=2)
    table[77, 5] = np.nan
    ids = np.array([1, 2, 3, 77, 4, 5, 6], dtype=np.uint32)
    offsets = np.array([0, 3, 5, 7], dtype=np.uint64)           # line 1 holds the damaged token
    model = smt.Model(gpu_ctx, table, normalize=True)
    c = smt.Corpus(gpu_ctx)
    msg = _refused(lambda: model.embed(ids, offsets, append_to=c))
    assert "first: row 1" in msg and c.rows == 0, msg
    emb, first = model.embed(ids[:3], offsets[:2], append_to=c)   # the healthy line alone goes in
    assert c.rows == 1 and first == 0
    c.close(); model.close()


def test_a_dealt_append_with_one_bad_row_leaves_every_shard_as_it_was(gpu_ctx):
    import semtools_amd as smt

    emb = synth.unit_rows(4000, seed=12)
    g = smt.Group.logical(0, 3)
    sc = smt.ShardedCorpus(g, empty=True)
    sc.append(emb[:1000])
    before = sc.rank_rows()
    bad = emb[1000:].copy()
    bad[2500, 3] = np.inf                                        # lands in the LAST shard's share: the others must roll back
    _refused(lambda: sc.append(bad))
    assert sc.rows == 1000 and sc.rank_rows().tolist() == before.tolist()
    sc.append(emb[1000:])
    q = synth.unit_query(8)[0]
    rows, dist = sc.search(q, top_k=4)[0]
    ref = orc.search_documents(emb, [4000], q, n_lines=0, top_k=4, accurate=True)
    assert rows.tolist() == [r["match_line"] for r in ref] and np.array_equal(dist, [r["distance"] for r in ref])
    _refused(lambda: smt.ShardedCorpus(g, rows=np.concatenate([emb, bad])))
    sc.close(); g.close()


def test_queries_outside_the_domain_are_refused_or_flagged(gpu_ctx):
    """Host forms: SMT_E_INVALID before anything is launched.  Device forms (nothing synchronises): SMT_STATUS_INVALID_QUERY for
    THAT query of THAT call, the others answered and proved as ever -- through the scan kernel (3 queries) and the batched one (16)."""
    import torch
    import semtools_amd as smt
    from semtools_amd import _lib as L

    emb = synth.unit_rows(30000, seed=13)
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    ix = smt.IvfPq(c, nlist=64, train_iters=4)
    g = smt.Group.logical(0, 2)
    sc = smt.ShardedCorpus(g, rows=emb)
    for kind in ("nan", "+inf", "huge", "tiny", "denormal"):
        for nq, at in ((1, 0), (3, 2), (16, 5)):
            qs = synth.unit_query(40 + nq, nq=nq)
            qs[at] = BAD_ROWS[kind](qs[at])
            for fn in (lambda: c.search(qs, top_k=5), lambda: c.search(qs, max_distance=0.5),
                       lambda: c.search(qs, top_k=5, mode=L.MODE_WORKSPACE, max_distance=0.9, ranges=[(0, 100)]),
                       lambda: c.search(qs, top_k=100), lambda: ix.search(qs, top_k=5, nprobe=4),
                       lambda: sc.search(qs, top_k=5)):
                assert f"query {at}" in _refused(fn)
            gpu_ctx.uncertain_count()
            qd = torch.from_numpy(qs).cuda()
            o_rows = torch.empty((nq, 5), dtype=torch.int64, device="cuda")
            o_dist = torch.empty((nq, 5), dtype=torch.float64, device="cuda")
            st = torch.full((nq,), 7, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            c.search_topk_device(qd.data_ptr(), nq, 5, 0, o_rows.data_ptr(), o_dist.data_ptr(), out_status_ptr=st.data_ptr())
            gpu_ctx.synchronize()
            want = [0] * nq
            want[at] = L.STATUS_INVALID_QUERY
            assert st.cpu().tolist() == want, (kind, nq, at)
            assert gpu_ctx.uncertain_count() == 1
            rows, dist = o_rows.cpu().numpy(), o_dist.cpu().numpy()
            for i in range(nq):
                if i != at:
                    ref = orc.search_documents(emb, [len(emb)], qs[i], n_lines=0, top_k=5, accurate=True)
                    assert rows[i].tolist() == [r["match_line"] for r in ref], (kind, nq, i)
                    assert np.array_equal(dist[i], [r["distance"] for r in ref])
    # the sharded device form carries the worst status of its shards
    qs = synth.unit_query(77, nq=4)
    qs[1, 9] = np.nan
    qd = torch.from_numpy(qs).cuda()
    packed = torch.zeros((4, 2, 5), dtype=torch.int64, device="cuda")
    st = torch.full((4,), 7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    sc.search_topk_device([qd.data_ptr(), qd.data_ptr()], 4, 5, [packed.data_ptr(), 0], out_status_ptrs=[st.data_ptr(), 0])
    g.synchronize()
    assert st.cpu().tolist() == [0, L.STATUS_INVALID_QUERY, 0, 0]
    sc.close(); g.close(); ix.close(); c.close()


# ------------------------------------------------------------------------------------------- inside the domain, at its edges
def _scaled_corpus(n, seed):
    """Unit rows (with duplicates and zero rows) each multiplied by its own power of two so that the largest magnitude of a row lands
    anywhere in [2^-40, 2^40] -- a few rows exactly at either end -- plus rows whose components span the whole range at once."""
    rng = np.random.default_rng(seed)
    base = synth.unit_rows(n, seed=seed, dup_frac=0.02, zero_frac=0.002)
    amax = np.abs(base).max(axis=1)
    nz = amax > 0
    # exponent of the largest magnitude after scaling, uniform over [-40, 39]; the scale is the power of two that puts it there
    e_max = np.floor(np.log2(np.where(nz, amax, 1.0))).astype(np.int64)
    target = rng.integers(-40, 40, n)
    target[:8] = -40
    target[8:16] = 39
    scale = np.exp2((target - e_max).astype(np.float64)).astype(np.float32)
    # rows with a huge and a tiny component side by side (the tiny ones vanish in f32 sums of squares: they must not matter): the
    # smallest component of a few rows becomes 2^-40 AFTER scaling -- written into `base` divided by the row's scale, so that
    # emb == base * scale stays exact
    wide = rng.choice(n, 32, replace=False)
    for r in wide:
        if nz[r] and target[r] >= 0:
            base[r, int(np.abs(base[r]).argmin())] = np.float32(2.0 ** -40) / scale[r]
    emb = base * scale[:, None]
    assert emb.dtype == np.float32 and np.array_equal(emb / scale[:, None], base)
    a = np.abs(emb).max(axis=1)
    assert ((a == 0) | ((a >= 2.0 ** -40) & (a <= 2.0 ** 40))).all()
    return base, emb


def _oracle_topk(emb, q, k):
    res = orc.search_documents(emb, [len(emb)], q, n_lines=0, top_k=k, accurate=True)
    return [r["match_line"] for r in res], np.array([r["distance"] for r in res])


@pytest.fixture(scope="module")
def scaled(gpu_ctx):
    import semtools_amd as smt

    base, emb = _scaled_corpus(40000, seed=21)
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    yield base, emb, c
    c.close()


def _queries(nq, seed):
    qs = synth.unit_query(seed, nq=nq)
    e = np.random.default_rng(seed).integers(-36, 41, nq)
    e[0] = -36
    if nq > 1:
        e[1] = 40
    return (qs * np.exp2(e.astype(np.float64)).astype(np.float32)[:, None]).astype(np.float32)


@pytest.mark.parametrize("nq", [1, 2, 3, 4])
def test_scan_kernel_at_the_edges_of_the_domain(gpu_ctx, scaled, nq):
    base, emb, c = scaled
    qs = _queries(nq, 30 + nq)
    gpu_ctx.set_tuning("gemm_min_nq", 8)                         # (keep 3 and 4 queries on the scan kernel)
    try:
        got = c.search(qs, top_k=10)
    finally:
        gpu_ctx.set_tuning("gemm_min_nq", 5)
    for i in range(nq):
        rows, dist = _oracle_topk(emb, qs[i], 10)
        assert got[i][0].tolist() == rows and np.array_equal(got[i][1], dist), i


@pytest.mark.parametrize("mode", [(1, 1, 1), (1, 1, 2), (1, 1, 3), (1, 0, 1), (0, 0, 1)],
                         ids=["bf16x3-rowreg", "f16x2-rowreg", "f16x1-rowreg", "bf16x3-level", "f32mfma"])
@pytest.mark.parametrize("image", [False, True], ids=["f32rows", "image"])
def test_batched_kernels_at_the_edges_of_the_domain(gpu_ctx, scaled, mode, image):
    base, emb, c = scaled
    bf16, rowreg, nominate = mode
    qs = _queries(40, 55)
    gpu_ctx.set_tuning("gemm_bf16x3", bf16)
    gpu_ctx.set_tuning("gemm_rowreg", rowreg)
    gpu_ctx.set_tuning("gemm_nominate", nominate)
    try:
        c.prepack(image)
        got = c.search(qs, top_k=10)
        few = c.search(qs[:6], top_k=3)                          # (5..7 queries: K3 as well)
    finally:
        c.prepack(False)
        gpu_ctx.set_tuning("gemm_bf16x3", 1)
        gpu_ctx.set_tuning("gemm_rowreg", 1)
        gpu_ctx.set_tuning("gemm_nominate", 0)
    for i in range(len(qs)):
        rows, dist = _oracle_topk(emb, qs[i], 10)
        assert got[i][0].tolist() == rows and np.array_equal(got[i][1], dist), i
    for i in range(6):
        rows, dist = _oracle_topk(emb, qs[i], 3)
        assert few[i][0].tolist() == rows and np.array_equal(few[i][1], dist), i


def test_threshold_large_k_ranges_and_workspace_mode_at_the_edges_of_the_domain(gpu_ctx, scaled):
    from semtools_amd import _lib as L

    base, emb, c = scaled
    qs = _queries(3, 71)
    q = qs[0]
    # K4: everything under a threshold (strict), in the reference's order
    ref = orc.search_documents(emb, [len(emb)], q, n_lines=0, top_k=0, max_distance=0.85, accurate=True)
    rows, dist = c.search(q, max_distance=0.85)[0]
    assert len(ref) > 50
    assert rows.tolist() == [r["match_line"] for r in ref] and np.array_equal(dist, [r["distance"] for r in ref])
    # large k (all keys + radix sort)
    rows, dist = c.search(qs[1], top_k=300)[0]
    orows, odist = _oracle_topk(emb, qs[1], 300)
    assert rows.tolist() == orows and np.array_equal(dist, odist)
    # range-filtered, documents and workspace mode (score > 1 - max_distance in f32), one query and a batch
    ranges = [(100, 1100), (5000, 5003), (20000, 33000)]
    keep = np.concatenate([np.arange(a, b) for a, b in ranges])
    for batch in (qs[2:3], _queries(12, 72)):
        got = c.search(batch, top_k=7, ranges=ranges)
        for i, qq in enumerate(batch):
            orows, odist = _oracle_topk(emb[keep], qq, 7)
            assert got[i][0].tolist() == keep[orows].tolist() and np.array_equal(got[i][1], odist), i
    row_path = np.zeros(len(emb), dtype=np.uint32)
    row_path[keep] = 1
    row_line = np.arange(len(emb), dtype=np.int32)
    # (the store's oracle is asked about the UNIT rows and the unit query: qdrant normalises vectors on insert -- except those with
    # |x|^2 < f32::EPSILON, which its cosine_preprocess leaves as they are, so that the restatement is not scale-invariant below
    # 3.4e-4 of length; the library applies cosine semantics to every in-domain row, which is what qdrant does to every row the
    # reference's own path can produce: unit or zero)
    got = c.search(qs[2], top_k=7, mode=L.MODE_WORKSPACE, max_distance=0.95, ranges=ranges)[0]
    unit_q = synth.unit_query(71, nq=3)[2]                       # (what _queries(3, 71) scaled by a power of two)
    ref = orc.search_line_embeddings(base, row_path, row_line, unit_q, [1], 7, 0.95)
    assert got[0].tolist() == [r["row"] for r in ref]
    assert np.allclose(got[1].astype(np.float32), [r["distance"] for r in ref], rtol=0, atol=1e-5)


def test_a_power_of_two_scale_changes_no_answer(gpu_ctx, scaled):
    """Cosine distance does not see the length of a vector, and a power-of-two scale is exact in every f32 and f64 operation of the
    path as long as nothing overflows or underflows -- which is what the domain guarantees.  The scaled corpus and the scaled
    queries must therefore give the rows AND the f64 distance bits the unit-length ones give: single queries, a batch from f32 rows
    and from the operand image, three logical shards, the approximate index."""
    import semtools_amd as smt

    base, emb, c = scaled
    unit = smt.Corpus(gpu_ctx)
    unit.append(base)
    qs_unit = synth.unit_query(91, nq=40)
    e = np.random.default_rng(91).integers(-36, 41, 40)
    qs = (qs_unit * np.exp2(e.astype(np.float64)).astype(np.float32)[:, None]).astype(np.float32)
    for image in (False, True):
        c.prepack(image); unit.prepack(image)
        for sel in (slice(0, 1), slice(0, 4), slice(0, 40)):
            a = c.search(qs[sel], top_k=10)
            b = unit.search(qs_unit[sel], top_k=10)
            for (ra, da), (rb, db) in zip(a, b):
                assert ra.tolist() == rb.tolist() and np.array_equal(da, db)
    c.prepack(False)
    g = smt.Group.logical(0, 3)
    sc = smt.ShardedCorpus(g, rows=emb)
    a = sc.search(qs[:5], top_k=10)
    b = unit.search(qs_unit[:5], top_k=10)
    for (ra, da), (rb, db) in zip(a, b):
        assert ra.tolist() == rb.tolist() and np.array_equal(da, db)
    sc.close(); g.close()
    # the approximate index ranks rows as they are, not their directions (its quantisers are fitted to unit rows): it refuses a corpus
    # holding rows of other lengths instead of answering with a recall nobody asked for; scaled QUERIES are fine
    from semtools_amd import _lib as L

    with pytest.raises(L.SmtError) as e:
        smt.IvfPq(c, nlist=64, train_iters=4, local_pca=True)
    assert e.value.code == -6 and "not unit-length" in str(e.value), e.value       # SMT_E_UNSUPPORTED
    ib = smt.IvfPq(unit, nlist=64, train_iters=6, local_pca=True)
    # (isotropic random rows have no cluster structure for probing to exploit: every list is probed and most of each re-scored, so
    # that what is compared is the index's arithmetic on scaled and unit queries, not its recall on data it was not made for)
    ra = ib.search(qs[:20], top_k=10, nprobe=64, rerank=512)
    rb = ib.search(qs_unit[:20], top_k=10, nprobe=64, rerank=512)
    exact = unit.search(qs_unit[:20], top_k=10)
    hit = sum(len(set(x[0].tolist()) & set(y[0].tolist())) for x, y in zip(ra, exact))
    assert hit >= 0.9 * 200, hit
    for (r1, d1), (r2, d2) in zip(ra, rb):
        assert r1.tolist() == r2.tolist() and np.array_equal(d1, d2)
    ib.close(); unit.close()
