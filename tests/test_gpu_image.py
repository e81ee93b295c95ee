"""The fp16 operand image of a corpus (smt_corpus_prepack, include/semtools_hip.h): derived data -- a batched search returns the
same rows and the same f64 distances with it and without it; appends, writes and truncation keep it current."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _unit(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, 256)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def _topk64(rows, q, k):
    d = 1.0 - (q.astype(np.float64) @ rows.astype(np.float64).T) / (
        np.linalg.norm(q.astype(np.float64), axis=1)[:, None] * np.maximum(np.linalg.norm(rows.astype(np.float64), axis=1), 1e-300)[None, :])
    idx = np.argsort(d, axis=1, kind="stable")[:, :k]
    return idx, np.take_along_axis(d, idx, axis=1)


@pytest.fixture(scope="module")
def smt():
    import semtools_amd as smt
    return smt


def _corpus(smt, ctx, rows):
    c = smt.Corpus(ctx)
    c.append(rows)
    return c


def _search(c, q, k):
    res = c.search(q, top_k=k)
    return [list(r) for r, _ in res], [list(d) for _, d in res]


@pytest.mark.parametrize("nq", [8, 40, 130, 300])
def test_same_answers_with_and_without_the_image(smt, nq):
    ctx = smt.Context(0)
    rows = _unit(200_003, 1)            # a ragged last tile
    rows[77] = 0.0                      # a zero row
    q = _unit(nq, 2)
    q[3] = 0.0                          # a zero query
    plain = _corpus(smt, ctx, rows)
    ctx.set_tuning("corpus_image", 0)
    r0, d0 = _search(plain, q, 10)
    assert plain.image_bytes == 0
    ctx.set_tuning("corpus_image", 1)
    packed = _corpus(smt, ctx, rows)
    r1, d1 = _search(packed, q, 10)
    assert packed.image_bytes >= (200_003 + 31) // 32 * 16384      # built by the first batch
    for a, b, da, db in zip(r0, r1, d0, d1):
        assert list(a) == list(b)
        assert list(da) == list(db)     # f64 re-scored from the f32 rows either way
    want_rows, want_d = _topk64(rows, q[[0, 5]], 10)
    for i, qi in enumerate((0, 5)):
        assert list(r1[qi]) == list(want_rows[i])
        np.testing.assert_allclose(d1[qi], want_d[i], atol=1e-12)


def test_image_follows_appends_writes_and_truncation(smt):
    ctx = smt.Context(0)
    rows = _unit(70_000, 3)
    q = _unit(16, 4)
    c = _corpus(smt, ctx, rows[:66_000])
    _search(c, q, 5)
    assert c.image_bytes > 0
    # append: the planted row must be found at once
    extra = rows[66_000:].copy()
    extra[1234] = q[7]
    c.append(extra)
    r, d = _search(c, q, 5)
    assert r[7][0] == 66_000 + 1234 and d[7][0] < 1e-6
    # write: the answer moves with the row
    c.write_rows(100, q[2:3])
    r, d = _search(c, q, 5)
    assert r[2][0] == 100 and d[2][0] < 1e-6
    # truncate below the planted rows, then append other rows into the same tiles
    c.truncate(66_010)
    c.append(rows[:500])
    r, d = _search(c, q, 5)
    assert r[7][0] != 66_000 + 1234
    full = np.concatenate([rows[:66_000], extra[:10], rows[:500]])
    full[100] = q[2]
    want_rows, want_d = _topk64(full, q, 5)
    for i in range(16):
        assert list(r[i]) == list(want_rows[i])
        np.testing.assert_allclose(d[i], want_d[i], atol=1e-12)


def test_adopted_rows_get_an_image_only_on_request(smt):
    ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    x = torch.from_numpy(_unit(100_000, 5)).cuda()
    q = _unit(64, 6)
    c = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=x.shape[0])
    r0, d0 = _search(c, q, 10)
    assert c.image_bytes == 0
    c.prepack()
    assert c.image_bytes > 0
    r1, d1 = _search(c, q, 10)
    for a, b, da, db in zip(r0, r1, d0, d1):
        assert list(a) == list(b) and list(da) == list(db)
    # the caller changed a row: prepack again
    x[4321] = torch.from_numpy(q[9]).cuda()
    torch.cuda.synchronize()
    c.prepack()
    r2, d2 = _search(c, q, 10)
    assert r2[9][0] == 4321 and d2[9][0] < 1e-6
    c.prepack(False)
    assert c.image_bytes == 0
    r3, d3 = _search(c, q, 10)
    assert list(r3[9]) == list(r2[9])


def test_dealt_shards_build_their_images_and_answer_alike(smt):
    """A corpus dealt over three shards of a logical group (what the host layer holds): every shard owns its rows, so each
    builds its own image at the first batch; answers equal the unsharded corpus', before and after an append and a write."""
    ctx = smt.Context(0)
    g = smt.Group.logical(0, 3)
    rows = _unit(400_000, 7)
    q = _unit(24, 8)
    one = _corpus(smt, ctx, rows[:300_000])
    sc = smt.ShardedCorpus(g, empty=True)
    sc.append(rows[:300_000])
    want_r, want_d = _search(one, q, 10)
    got = sc.search(q, top_k=10)
    assert [list(r) for r, _ in got] == want_r and [list(d) for _, d in got] == want_d
    assert all(sc.shard(i, want_base=False)[0].image_bytes > 0 for i in range(3))
    extra = rows[300_000:].copy()
    extra[777] = q[5]
    sc.append(extra)
    one.append(extra)
    sc.write_rows(123_456, q[6:7])
    one.write_rows(123_456, q[6:7])
    want_r, want_d = _search(one, q, 10)
    got = sc.search(q, top_k=10)
    assert [list(r) for r, _ in got] == want_r and [list(d) for _, d in got] == want_d
    assert got[5][0][0] == 300_777 and got[6][0][0] == 123_456
    sc.close()
    g.close()


def test_one_query_scans_the_image_of_a_large_shard(smt):
    """topk_dispatch: a shard of >= image_scan_min_rows rows (1.5 M by default; lowered here) that HAS its image answers one or two
    queries through the batched kernel (512 B per row) -- same rows, same f64 distances as the scan kernel over the f32 rows."""
    ctx = smt.Context(0)
    rows = _unit(300_000, 9)
    q = _unit(2, 10)
    rows[[5, 299_999]] = q[0]                      # a tie across the whole corpus: row order decides
    c = _corpus(smt, ctx, rows)
    want = [_search(c, q[:n], 10) for n in (1, 2)]  # scan kernel (no image yet: fewer than 8 queries never build one)
    assert c.image_bytes == 0
    c.prepack()
    ctx.set_tuning("image_scan_min_rows", 100_000)
    try:
        for n in (1, 2):
            ctx.prof_enable(True)
            ctx.prof_reset()
            got = _search(c, q[:n], 10)
            launches, _ = ctx.prof_read("gemm")
            ctx.prof_enable(False)
            assert launches > 0                    # the batched kernel ran
            assert got == want[n - 1]
        assert want[0][0][0][:2] == [5, 299_999]
        ctx.set_tuning("image_scan_min_rows", 0)   # off: back on the scan kernel
        ctx.prof_enable(True)
        ctx.prof_reset()
        assert _search(c, q[:1], 10) == want[0]
        assert ctx.prof_read("gemm")[0] == 0
        ctx.prof_enable(False)
    finally:
        ctx.set_tuning("image_scan_min_rows", 1_500_000)


def test_a_resident_host_gets_the_image_after_a_few_single_queries(smt):
    """`semtools serve` asks one query at a time: the fourth small search of an owned shard that is large enough to scan its
    image builds it (topk_dispatch)."""
    ctx = smt.Context(0)
    rows = _unit(200_000, 11)
    q = _unit(6, 12)
    c = _corpus(smt, ctx, rows)
    ctx.set_tuning("image_scan_min_rows", 100_000)
    try:
        got = []
        for i in range(3):
            got.append(_search(c, q[i:i + 1], 10))
            assert c.image_bytes == 0
        got.append(_search(c, q[3:4], 10))
        assert c.image_bytes > 0
        ctx.set_tuning("image_scan_min_rows", 0)
        for i in range(4):
            assert _search(c, q[i:i + 1], 10) == got[i]          # the scan kernel says the same
    finally:
        ctx.set_tuning("image_scan_min_rows", 1_500_000)


def test_random_life_of_a_corpus_with_an_image(smt):
    """Appends, overwrites, truncations and searches in random order on a corpus that keeps its image: every batch agrees with a
    brute-force f64 top-k over a numpy mirror of the rows (indices and distances)."""
    rng = np.random.default_rng(20)
    ctx = smt.Context(0)
    mirror = _unit(66_000, 21)
    c = _corpus(smt, ctx, mirror)
    c.prepack()
    pool = _unit(40_000, 22)
    for step in range(24):
        op = rng.integers(0, 4)
        if op == 0:                                   # append 1 .. 3000 rows (crosses tile borders at random places)
            n = int(rng.integers(1, 3000))
            rows = pool[rng.integers(0, len(pool), n)]
            c.append(rows)
            mirror = np.concatenate([mirror, rows])
        elif op == 1:                                 # overwrite a run somewhere
            n = int(rng.integers(1, 200))
            at = int(rng.integers(0, len(mirror) - n))
            rows = pool[rng.integers(0, len(pool), n)]
            c.write_rows(at, rows)
            mirror[at:at + n] = rows
        elif op == 2 and len(mirror) > 66_000:        # cut the tail, sometimes inside a tile
            to = int(rng.integers(65_600, len(mirror)))
            c.truncate(to)
            mirror = mirror[:to]
        q = _unit(8, 100 + step)
        q[0] = mirror[int(rng.integers(0, len(mirror)))]          # a query that sits in the corpus (duplicates tie by row order)
        got_r, got_d = _search(c, q, 7)
        want_r, want_d = _topk64(mirror, q, 7)
        for i in range(8):
            assert got_r[i] == list(want_r[i]), (step, i)
            np.testing.assert_allclose(got_d[i], want_d[i], atol=1e-12)
        assert c.image_bytes > 0


@pytest.mark.parametrize("nominate", [0, 2, 3])
def test_near_tie_clusters_over_the_image_stay_exact(smt, nominate):
    """Forty rows within 1e-4 of every query -- closer together than the fp16 nomination can tell apart -- plus exact duplicates:
    the certificate fails, the exhaustive re-answer takes over, and the answers over the operand image are the f64 truth in every
    fp16 mode (gemm_nominate 2 / 3 forced, 0 = the default routing)."""
    rng = np.random.default_rng(30)
    ctx = smt.Context(0)
    rows = _unit(90_000, 31)
    q = _unit(12, 32)
    for i in range(12):
        at = rng.choice(90_000, 42, replace=False)
        near = q[i][None, :] + 1e-4 * rng.standard_normal((42, 256)).astype(np.float32)
        rows[at] = near / np.linalg.norm(near, axis=1, keepdims=True)
        rows[at[:2]] = rows[at[2]]                    # three identical rows: row order decides
    c = _corpus(smt, ctx, rows)
    c.prepack()
    ctx.set_tuning("gemm_nominate", nominate)
    try:
        got_r, got_d = _search(c, q, 10)
    finally:
        ctx.set_tuning("gemm_nominate", 0)
    want_r, want_d = _topk64(rows, q, 10)
    for i in range(12):
        assert got_r[i] == list(want_r[i]), i
        np.testing.assert_allclose(got_d[i], want_d[i], atol=1e-12)
