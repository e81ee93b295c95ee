"""GPU: IVF-PQ index (BASELINE config 5).  No reference semantics exist, so the checks are:
recall@k against this library's own exact search on clustered data, exactness of every returned
(row, distance) pair, monotone recall in nprobe, determinism of build + query."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def clustered(n, n_centers, seed, latent=8, spread=0.35, noise=0.01):
    """Clustered corpus with graded neighbourhoods (config c5's 'clustered variant'): row = topic centre +
    spread * z . B_topic + noise, normalised, z ~ N(0, I_latent), B_topic a per-topic latent x 256 basis.
    Nearest neighbours of a row = rows of its topic that are close in the latent space: a well-defined top-k
    with structure a quantiser can encode (isotropic 256-d noise has none)."""
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((n_centers, 256)).astype(np.float32)
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    basis = rng.standard_normal((n_centers, latent, 256)).astype(np.float32) / 16.0
    which = rng.integers(0, n_centers, n)
    z = rng.standard_normal((n, latent)).astype(np.float32) * (spread / np.sqrt(latent))
    x = centers[which] + np.einsum("nl,nld->nd", z, basis[which]) + (noise / 16.0) * rng.standard_normal((n, 256)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x.astype(np.float32)), centers


@pytest.fixture(scope="module")
def index(gpu_ctx):
    import semtools_amd as smt

    x, centers = clustered(60000, 256, seed=1)
    c = smt.Corpus(gpu_ctx)
    c.append(x)
    ix = smt.IvfPq(c, nlist=256, train_iters=8)
    yield x, c, ix
    ix.close(); c.close()


def recall(ix_res, exact_res):
    hit = tot = 0
    for (r, _), (e, _) in zip(ix_res, exact_res):
        hit += len(set(r.tolist()) & set(e.tolist()))
        tot += len(e)
    return hit / tot


def test_build_partitions_every_row_once(index):
    x, c, ix = index
    info = ix.info()
    sizes = ix.list_sizes()
    assert info["rows"] == len(x) and info["nlist"] == 256 and int(sizes.sum()) == len(x)
    assert sizes.max() < 20 * sizes.mean()                     # k-means on blobs: no giant list
    assert info["index_bytes"] < len(x) * 40 + 2_000_000       # 32 B codes + 4 B ids per row + codebooks


def test_recall_and_exact_distances(index):
    x, c, ix = index
    rng = np.random.default_rng(5)
    qs = x[rng.choice(len(x), 64, replace=False)] + 0.002 * rng.standard_normal((64, 256)).astype(np.float32)
    exact = c.search(qs, top_k=10)
    prev = 0.0
    for nprobe in (1, 4, 16, 64):
        got = ix.search(qs, top_k=10, nprobe=nprobe)
        r = recall(got, exact)
        assert r >= prev - 0.02, (nprobe, r, prev)              # more lists never hurts (up to ADC noise)
        prev = r
        for qi, (rows, dist) in enumerate(got):                  # every returned pair is an exact pair
            want = np.array([orc.cosine(qs[qi], x[int(rr)], accurate=True) for rr in rows])
            assert np.array_equal(dist, want) and (np.diff(dist) >= 0).all() and len(set(rows.tolist())) == len(rows)
    assert prev >= 0.97, prev                                    # nprobe=64 of 256 lists, default rerank (256/list)
    shallow = ix.search(qs, top_k=10, nprobe=64, rerank=16)     # fewer re-scored candidates: recall may only drop
    assert recall(shallow, exact) <= prev + 1e-9


def test_self_queries_are_found_first(index):
    x, c, ix = index
    ids = [0, 123, 59999, 31415]
    got = ix.search(x[ids], top_k=3, nprobe=8)
    for i, (rows, dist) in zip(ids, got):
        assert dist[0] < 1e-12 and int(rows[0]) in np.nonzero((x == x[i]).all(axis=1))[0].tolist()


def test_build_and_search_are_deterministic(gpu_ctx):
    import semtools_amd as smt

    x, _ = clustered(20000, 64, seed=2)
    c = smt.Corpus(gpu_ctx)
    c.append(x)
    a = smt.IvfPq(c, nlist=64, train_iters=5)
    b = smt.IvfPq(c, nlist=64, train_iters=5)
    assert a.list_sizes().tolist() == b.list_sizes().tolist()   # fixed-point accumulation: order independent
    q = x[:16] + 0.01
    ra, rb = a.search(q, top_k=5, nprobe=4), b.search(q, top_k=5, nprobe=4)
    for (r1, d1), (r2, d2) in zip(ra, rb):
        assert r1.tolist() == r2.tolist() and np.array_equal(d1, d2)
    a.close(); b.close(); c.close()


def test_argument_validation(index):
    import semtools_amd as smt

    x, c, ix = index
    with pytest.raises(smt.SmtError):
        smt.IvfPq(c, nlist=100)             # not a multiple of 32
    with pytest.raises(smt.SmtError):
        ix.search(x[:1], top_k=10, nprobe=0)
    with pytest.raises(smt.SmtError):
        ix.search(x[:1], top_k=57, nprobe=4)
    assert ix.search(x[:1], top_k=0, nprobe=4)[0][0].size == 0


def test_save_load_round_trip(index, gpu_ctx, tmp_path):
    """A restored index answers exactly like the one that was saved; a file that does not match the corpus
    (row count) or is damaged is refused, never searched."""
    import semtools_amd as smt

    x, c, ix = index
    path = tmp_path / "lines.ivfpq"
    ix.save(path)
    assert path.stat().st_size == 64 + 256 * 1024 + 256 * 4 + 32 * 256 * 8 * 4 + 257 * 8 + len(x) * 36
    back = smt.IvfPq.load(c, path)
    assert back.info()["rows"] == len(x) and np.array_equal(back.list_sizes(), ix.list_sizes())
    qs = x[[5, 999, 42424]] + np.float32(0.001)
    for (r1, d1), (r2, d2) in zip(ix.search(qs, top_k=10, nprobe=8), back.search(qs, top_k=10, nprobe=8)):
        assert r1.tolist() == r2.tolist() and np.array_equal(d1, d2)
    back.close()

    other = smt.Corpus(gpu_ctx)
    other.append(x[:1000])
    with pytest.raises(smt.SmtError, match="rebuild"):
        smt.IvfPq.load(other, path)
    other.close()
    blob = path.read_bytes()
    (tmp_path / "short.ivfpq").write_bytes(blob[: len(blob) // 2])
    with pytest.raises(smt.SmtError):
        smt.IvfPq.load(c, tmp_path / "short.ivfpq")
    (tmp_path / "magic.ivfpq").write_bytes(b"NOTANIDX" + blob[8:])
    with pytest.raises(smt.SmtError):
        smt.IvfPq.load(c, tmp_path / "magic.ivfpq")
    with pytest.raises(smt.SmtError):
        smt.IvfPq.load(c, tmp_path / "missing.ivfpq")


def test_device_resident_search_equals_host_search(index):
    import torch

    x, c, ix = index
    qs = x[[3, 1000, 20000, 59999]] + np.float32(0.001)
    want = ix.search(qs, top_k=7, nprobe=8, row_base=12345)
    qd = torch.from_numpy(np.ascontiguousarray(qs)).cuda()
    rows = torch.empty((4, 7), dtype=torch.int64, device="cuda")
    dist = torch.empty((4, 7), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ix.search_device(qd.data_ptr(), 4, 7, 8, 0, 12345, rows.data_ptr(), dist.data_ptr())
    c.ctx.synchronize()
    r = rows.cpu().numpy().view(np.uint64)
    d = dist.cpu().numpy()
    for i, (wr, wd) in enumerate(want):
        assert r[i, : len(wr)].tolist() == wr.tolist() and np.array_equal(d[i, : len(wd)], wd)


def test_every_returned_distance_is_the_exact_one_of_its_row(gpu_ctx):
    """Approximate membership, exact values: whatever the ADC shortlist nominates, a returned (row, distance) pair carries
    the f64 distance the exact search gives that row."""
    import semtools_amd as smt

    x, _ = clustered(40000, 128, seed=3)
    c = smt.Corpus(gpu_ctx)
    c.append(x)
    rng = np.random.default_rng(6)
    qs = x[rng.choice(len(x), 200, replace=False)] + 0.002 * rng.standard_normal((200, 256)).astype(np.float32)
    exact = c.search(qs, top_k=10)
    for local_pca in (False, True):
        ix = smt.IvfPq(c, nlist=128, train_iters=6, local_pca=local_pca)
        assert ix.info()["index_bytes"] < len(x) * 36 + (8 << 20)          # 36 B per row + centroids / codebooks / per-list bases
        got = ix.search(qs, top_k=10, nprobe=8)
        recall = sum(len(set(r.tolist()) & set(e.tolist())) for (r, _), (e, _) in zip(got, exact)) / 2000
        assert recall > 0.9, (local_pca, recall)
        for (r, d), (er, ed) in zip(got, exact):
            lut = dict(zip(er.tolist(), ed.tolist()))
            assert all(abs(dd - lut[rr]) == 0.0 for rr, dd in zip(r.tolist(), d.tolist()) if rr in lut)
        ix.close()
    c.close()


def test_long_lists_are_scanned_in_segments(gpu_ctx):
    """Lists longer than 8192 codes (here: 200 k rows over 32 lists, 6250 per list => two segments per probed list, the
    second one partial, and any longer list spilling into the last segment): results keep the exact distances, good
    recall, and the device-resident entry point agrees with the host one."""
    import torch
    import semtools_amd as smt

    from tests import synth

    x = synth.clustered_rows_torch(200_000, 32, 8, 21, "cuda")
    torch.cuda.synchronize()
    c = smt.Corpus(gpu_ctx, device_ptr=x.data_ptr(), rows=200_000)
    ix = smt.IvfPq(c, nlist=32, train_iters=5)
    assert ix.list_sizes().mean() * 1.5 > 8192
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    qi = torch.randint(0, 200_000, (64,), device="cuda", generator=g)
    qs = (x[qi] + 0.002 * torch.randn(64, 256, device="cuda", generator=g)).cpu().numpy()
    exact = c.search(qs, top_k=10)
    got = ix.search(qs, top_k=10, nprobe=4)
    again = ix.search(qs, top_k=10, nprobe=4)
    hit = 0
    for (r, d), (r2, d2), (er, ed) in zip(got, again, exact):
        assert r.tolist() == r2.tolist() and np.array_equal(d, d2)           # deterministic
        lut = dict(zip(er.tolist(), ed.tolist()))
        assert all(dd == lut[rr] for rr, dd in zip(r.tolist(), d.tolist()) if rr in lut)
        hit += len(set(r.tolist()) & set(er.tolist()))
    assert hit / 640 > 0.85, hit / 640
    ix.close()
    c.close()


def test_per_list_pca_codes(gpu_ctx, tmp_path):
    """Index kind 1 (per-list PCA basis + 8-bit scalar codes): better recall than the global codebooks at a small
    re-score depth on data whose topics do not coincide with the lists; exact (row, distance) pairs; deterministic
    build; save / load round trip."""
    import semtools_amd as smt

    x, _ = clustered(80000, 900, seed=4)                       # 900 topics over 256 lists
    rng = np.random.default_rng(6)
    qs, _ = clustered(64, 900, seed=4)                         # (same topic model: seed) ...
    qs = qs[rng.permutation(64)] + 0.0 * qs                     # ... rows of a FRESH draw, not corpus rows
    c = smt.Corpus(gpu_ctx)
    c.append(x)
    exact = c.search(qs, top_k=10)
    ix0 = smt.IvfPq(c, nlist=256, train_iters=8)
    ix1 = smt.IvfPq(c, nlist=256, train_iters=8, local_pca=True)
    r0 = recall(ix0.search(qs, top_k=10, nprobe=16, rerank=32), exact)
    got = ix1.search(qs, top_k=10, nprobe=16, rerank=32)
    r1 = recall(got, exact)
    assert r1 >= 0.9 and r1 >= r0, (r0, r1)
    for (rows, dist), q in zip(got, qs):                       # every returned pair is exact
        ref = np.array([orc.cosine(q, x[int(r)], accurate=True) for r in rows])
        assert np.array_equal(dist, ref) and (np.diff(dist) >= 0).all()
    ix1b = smt.IvfPq(c, nlist=256, train_iters=8, local_pca=True)
    again = ix1b.search(qs, top_k=10, nprobe=16, rerank=32)
    assert all(a[0].tolist() == b[0].tolist() for a, b in zip(got, again))          # deterministic build
    path = tmp_path / "lpca.ivf"
    ix1.save(path)
    ix1c = smt.IvfPq.load(c, path)
    loaded = ix1c.search(qs, top_k=10, nprobe=16, rerank=32)
    assert all(a[0].tolist() == b[0].tolist() and np.array_equal(a[1], b[1]) for a, b in zip(got, loaded))
    assert ix1c.info()["index_bytes"] == ix1.info()["index_bytes"] > ix0.info()["index_bytes"]
    for i in (ix0, ix1, ix1b, ix1c):
        i.close()
    c.close()


@pytest.mark.parametrize("local_pca", [False, True])
def test_incremental_append_equals_a_rebuild_on_the_same_quantisers(gpu_ctx, local_pca):
    """smt_ivfpq_append: rows added after the build are searchable, the old rows keep their codes, and the merged
    lists partition every row exactly once."""
    import semtools_amd as smt

    x, _ = clustered(50000, 300, seed=9)
    c = smt.Corpus(gpu_ctx)
    c.append(x[:40000])
    ix = smt.IvfPq(c, nlist=128, train_iters=6, local_pca=local_pca)
    c.append(x[40000:])
    assert ix.append() == 10000 and ix.append() == 0
    assert ix.info()["rows"] == 50000 and int(ix.list_sizes().sum()) == 50000
    rng = np.random.default_rng(3)
    qs = x[rng.choice(np.arange(40000, 50000), 32, replace=False)]          # queries that ARE new rows
    got = ix.search(qs, top_k=5, nprobe=128, rerank=512)
    exact = c.search(qs, top_k=5)
    assert recall(got, exact) >= 0.99
    for (rows, dist), (er, ed) in zip(got, exact):
        assert rows[0] == er[0] and dist[0] == ed[0]                          # each query finds itself (a new row)
    ix.close(); c.close()
