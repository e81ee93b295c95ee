"""GPU parity: K3 batched path (Q x C^T on the MFMA pipes + fused candidate selection) vs the oracle and
vs the single-query K2 path.  Same bar: indices exact, distances = oracle f64 values.  Every test runs four times:
candidates nominated in the row-register kernel by bf16 x 3 split products and by f16 x 2 (the library picks between
them by batch size; both are forced here), by bf16 x 3 in the round-2 kernels (gemm_rowreg = 0), and by f32 MFMAs
(gemm_bf16x3 = 0)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import synth
from tests.compare import assert_topk_tie_aware, reference_distances

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[(1, 1, 1), (1, 1, 2), (1, 1, 3), (1, 0, 1), (0, 0, 1)],
                ids=["bf16x3-rowreg", "f16x2-rowreg", "f16x1-rowreg", "bf16x3-level", "f32mfma"])
def nominate_with(request, gpu_ctx):
    bf16, rowreg, nominate = request.param
    gpu_ctx.set_tuning("gemm_bf16x3", bf16)
    gpu_ctx.set_tuning("gemm_rowreg", rowreg)
    gpu_ctx.set_tuning("gemm_nominate", nominate)
    gpu_ctx._rowreg_mode = bool(rowreg)
    gpu_ctx._f16x2_mode = bool(bf16 and rowreg and nominate == 2)
    gpu_ctx._f16x1_mode = bool(bf16 and rowreg and nominate == 3)
    # the bound compiled into the library for this mode (common.h F32_ERR_*)
    gpu_ctx._nominating_bound = (1.0e-3 if gpu_ctx._f16x1_mode else 5.2e-4 if gpu_ctx._f16x2_mode else 7e-5 if bf16 else 2e-5)
    yield bf16
    gpu_ctx.set_tuning("gemm_bf16x3", 1)
    gpu_ctx.set_tuning("gemm_rowreg", 1)
    gpu_ctx.set_tuning("gemm_nominate", 0)


def _oracle_topk(emb, q, k):
    res = orc.search_documents(emb, [len(emb)], q, n_lines=0, top_k=k, accurate=True)
    return [r["match_line"] for r in res], [r["distance"] for r in res]


@pytest.mark.parametrize("n_rows,nq,k", [(20000, 8, 5), (20000, 33, 10), (4097, 64, 3), (50000, 100, 10),
                                          (1000, 9, 24), (31, 8, 4), (70000, 40, 1), (30001, 200, 10), (5000, 300, 3)])
def test_batched_matches_oracle(gpu_ctx, n_rows, nq, k):
    import semtools_amd as smt

    emb = synth.unit_rows(n_rows, seed=3 + n_rows)
    qs = synth.unit_query(50 + nq, nq=nq)
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    got = c.search(qs, top_k=k)
    for i in range(nq):
        orows, odist = _oracle_topk(emb, qs[i], k)
        assert got[i][0].tolist() == orows, (i, got[i][0].tolist(), orows)
        assert np.array_equal(got[i][1], np.array(odist)), i
        if i < 4:   # tie-aware contract against the serial-f32 restatement
            assert_topk_tie_aware(got[i][0], got[i][1], reference_distances(emb, qs[i]), k)
    c.close()


def test_batched_with_ties_zero_rows_and_zero_query(gpu_ctx):
    import semtools_amd as smt

    emb = synth.unit_rows(10000, seed=77, dup_frac=0.1, zero_frac=0.01)
    qs = synth.unit_query(5, nq=16)
    qs[3] = 0.0                       # zero query: zero rows rank first (distance 0), the rest are 1
    qs[7] = emb[1234]                 # exact hit(s)
    emb[[11, 5000, 9999]] = qs[9]     # three exact copies of query 9 -> rows in ascending order
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    got = c.search(qs, top_k=6)
    for i in range(16):
        orows, odist = _oracle_topk(emb, qs[i], 6)
        assert got[i][0].tolist() == orows, i
        assert np.array_equal(got[i][1], np.array(odist)), i
    assert got[9][0][:3].tolist() == [11, 5000, 9999]
    c.close()


def test_batched_equals_single_query_path(gpu_ctx):
    import semtools_amd as smt

    emb = synth.unit_rows(30000, seed=5)
    qs = synth.unit_query(6, nq=12)
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    batch = c.search(qs, top_k=10)
    for i in range(12):
        one = c.search(qs[i], top_k=10)[0]
        assert batch[i][0].tolist() == one[0].tolist() and np.array_equal(batch[i][1], one[1])
    c.close()


def test_batched_sorted_corpus_is_still_exact(gpu_ctx):
    """Adversarial row order for the level thresholds: rows sorted by similarity to the queries'
    mean direction, so sampled tiles are unrepresentative.  Whatever path is taken (candidate
    buffers or the overflow fallback), results must stay exact."""
    import semtools_amd as smt

    emb = synth.unit_rows(40000, seed=9, dup_frac=0, zero_frac=0)
    qs = synth.unit_query(10, nq=8)
    order = np.argsort(emb @ qs.mean(axis=0))          # worst first, best last
    emb = np.ascontiguousarray(emb[order])
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    got = c.search(qs, top_k=10)
    for i in range(8):
        orows, odist = _oracle_topk(emb, qs[i], 10)
        assert got[i][0].tolist() == orows, i
    c.close()


def _flooding_corpus(gpu_ctx, stride):
    """Adversarial order for the level thresholds: every tile the bootstrap level samples (tile index = 0 mod its stride: 2 for a
    corpus of 2188 tiles since round 5, 16 with tuning key gemm_boot_fine = 0) holds far rows, everything else is near the queries."""
    gpu_ctx.set_tuning("gemm_boot_fine", 1 if stride == 2 else 0)
    n = 70_000
    rng = np.random.default_rng(3)
    qs = synth.unit_query(8, nq=9)
    far = synth.unit_rows(n, seed=4, dup_frac=0, zero_frac=0)
    near = qs[rng.integers(0, 9, n)] + 0.05 * rng.standard_normal((n, 256)).astype(np.float32)
    near /= np.linalg.norm(near, axis=1, keepdims=True)
    tile = np.arange(n) // 32
    return n, qs, np.where((tile % stride == 0)[:, None], far, near).astype(np.float32)


@pytest.mark.parametrize("stride", [2, 16])
def test_candidate_buffer_overflow_falls_back_to_exact_scan(gpu_ctx, stride):
    """The main level floods the 2048-slot candidate buffers (_flooding_corpus).
    The overflow flag must route those queries through the exact K2 scan: results stay exact."""
    import semtools_amd as smt

    try:
        n, qs, emb = _flooding_corpus(gpu_ctx, stride)
        c = smt.Corpus(gpu_ctx)
        c.append(emb)
        gpu_ctx.uncertain_count()
        got = c.search(qs, top_k=10)
        assert gpu_ctx.uncertain_count() >= 1          # (the corpus does flood: somebody was re-answered)
        for i in range(9):
            res = orc.search_documents(emb, [n], qs[i], 0, 10, accurate=True)
            assert got[i][0].tolist() == [r["match_line"] for r in res], i
            assert np.array_equal(got[i][1], np.array([r["distance"] for r in res]))
        c.close()
    finally:
        gpu_ctx.set_tuning("gemm_boot_fine", 1)


@pytest.mark.parametrize("nq", [8, 33, 70])
def test_batched_with_row_ranges_reaches_the_mfma_path(gpu_ctx, nq):
    """Range-filtered batches (the workspace path-subset filter with several queries, store.rs:507-515) run on the
    LDS-row MFMA kernel through the chunk-descriptor table: same answer as the oracle on the eligible rows and as
    the single-query (K2) path."""
    import semtools_amd as smt

    n = 30011
    emb = synth.unit_rows(n, seed=19)
    qs = synth.unit_query(21, nq=nq)
    ranges = [(3, 1001), (1002, 1003), (5000, 5002), (7777, 20000), (29990, 30011)]   # ragged: lengths 998, 1, 2, 12223, 21
    elig = np.concatenate([np.arange(b, e) for b, e in ranges])
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    for k in (1, 10):
        got = c.search(qs, top_k=k, ranges=ranges)
        for i in range(nq):
            orows, odist = _oracle_topk(emb[elig], qs[i], k)
            assert got[i][0].tolist() == elig[np.array(orows, dtype=np.int64)].tolist(), (k, i)
            assert np.array_equal(got[i][1], np.array(odist))
        one = c.search(qs[0], top_k=k, ranges=ranges)[0]
        assert got[0][0].tolist() == one[0].tolist() and np.array_equal(got[0][1], one[1])
    # workspace semantics over ranges, batched
    got = c.search(qs, top_k=5, max_distance=0.95, mode=smt.MODE_WORKSPACE, ranges=ranges)
    for i in (0, nq - 1):
        one = c.search(qs[i], top_k=5, max_distance=0.95, mode=smt.MODE_WORKSPACE, ranges=ranges)[0]
        assert got[i][0].tolist() == one[0].tolist()
    c.close()


@pytest.mark.parametrize("n_rows", [33 * 32 + 5, 64 * 32, 1500 * 32 + 1, 2049 * 32 + 7, 300_000])
def test_bootstrap_plan_and_appended_levels_plan_agree(gpu_ctx, n_rows):
    """The row-register kernel's level plan starts from a BOOTSTRAP level (tile minima, no candidate lists: gemm_topk.hip) instead
    of up to three appended levels.  Same answers as the round-1..3 plan (tuning key gemm_bootstrap = 0) and as the oracle, at the
    sizes where the plan changes shape: 34 tiles (the smallest bootstrap), 64, 1501 (two tiles folded per slot), 2050 (the first
    strided bootstrap), 9376; with a zero query, a zero row in the ragged last tile and duplicate rows."""
    import semtools_amd as smt

    emb = synth.unit_rows(n_rows, seed=7 + n_rows % 97)
    emb[-1] = 0.0
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    try:
        for nq in (8, 40, 300):
            qs = synth.unit_query(90 + nq, nq=nq)
            qs[3] = 0.0
            for k in (1, 10, 40):
                new = c.search(qs, top_k=k)
                gpu_ctx.set_tuning("gemm_bootstrap", 0)
                try:
                    old = c.search(qs, top_k=k)
                finally:
                    gpu_ctx.set_tuning("gemm_bootstrap", 1)
                for i, (x, y) in enumerate(zip(new, old)):
                    assert x[0].tolist() == y[0].tolist() and np.array_equal(x[1], y[1]), (nq, k, i)
                for i in (0, 3, nq - 1):
                    orows, odist = _oracle_topk(emb, qs[i], k)
                    assert new[i][0].tolist() == orows and np.array_equal(new[i][1], np.array(odist)), (nq, k, i)
    finally:
        c.close()


def test_level_kernel_and_lds_row_kernel_agree(gpu_ctx):
    """In the non-row-register modes up to 64 queries take the LDS-row kernel; gemm_ldsrow = 0 sends them through
    gemm_level_kernel instead (A/B runs).  Same answers."""
    import semtools_amd as smt

    emb = synth.unit_rows(40000, seed=23)
    qs = synth.unit_query(24, nq=100)
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    new = c.search(qs, top_k=10)
    gpu_ctx.set_tuning("gemm_ldsrow", 0)
    try:
        old = c.search(qs, top_k=10)
    finally:
        gpu_ctx.set_tuning("gemm_ldsrow", 1)
    for a, b in zip(new, old):
        assert a[0].tolist() == b[0].tolist() and np.array_equal(a[1], b[1])
    c.close()


def test_nominating_distance_error_is_inside_the_certificate_bound(gpu_ctx, nominate_with):
    """The certificate needs |nominating f32 distance - exact distance| <= the bound compiled into
    the library: 2e-5 for the f32 MFMA chain, 7e-5 for bf16 x 3, 5.2e-4 for f16 x 2, 1.0e-3 for f16 x 1.  Measured here on the corpora that stress it:
    isotropic rows, all-positive rows (sum |x_i q_i| = x.q: no cancellation, the largest accumulations), rows with
    a few dominant components, unnormalised rows and queries."""
    import semtools_amd as smt

    rng = np.random.default_rng(5)
    n = 4096
    iso = synth.unit_rows(n, seed=77, dup_frac=0, zero_frac=0)
    pos = np.abs(rng.standard_normal((n, 256))).astype(np.float32)
    spiky = (rng.standard_normal((n, 256)) * np.exp(3.0 * rng.standard_normal((n, 256)))).astype(np.float32)
    scaled = (rng.standard_normal((n, 256)) * 37.5).astype(np.float32)
    emb = np.ascontiguousarray(np.concatenate([iso, pos, spiky, scaled]))
    qs = np.concatenate([synth.unit_query(9, nq=8), np.abs(rng.standard_normal((8, 256))).astype(np.float32),
                         (rng.standard_normal((8, 256)) * np.exp(3.0 * rng.standard_normal((8, 256)))).astype(np.float32),
                         (rng.standard_normal((8, 256)) * 1e-3).astype(np.float32)])
    qs = np.ascontiguousarray(qs, dtype=np.float32)
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    got = c.debug_batched_scores(qs).astype(np.float64)                   # [rows, 32]
    e64, q64 = emb.astype(np.float64), qs.astype(np.float64)
    cos = (e64 @ q64.T) / (np.linalg.norm(e64, axis=1)[:, None] * np.linalg.norm(q64, axis=1)[None, :])
    exact = np.maximum(1.0 - cos, 0.0)
    err = np.abs(got - exact).max()
    fp16 = getattr(gpu_ctx, "_f16x2_mode", False) or getattr(gpu_ctx, "_f16x1_mode", False)
    bound = gpu_ctx._nominating_bound
    print(f"max |nominating - exact| = {err:.3e} (bound {bound:.1e})")
    # the compiled-in bounds are worst cases: bf16 x 3 and f32 sit far inside theirs (residual terms of random sign);
    # the fp16 modes' bound is the operands' rounding itself, which all-positive rows come within 2.5 x of
    assert err < (bound * 0.6 if fp16 else bound / 3), err
    c.close()


def _bf16_rne(x):
    """f32 -> bf16 (round to nearest even) -> f32, as v_cvt_pk_bf16_f32 does"""
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
    return b.astype(np.uint32).view(np.float32)


def _worst_case_rows_f16x2(rng, n_rows):
    """Unit rows whose elements x 2^10 sit EXACTLY on fp16 rounding midpoints, in two binades so that the row has unit norm
    to within 2^-14: the kernel's normalisation then shifts every element off its midpoint in the SAME direction and all
    256 roundings go the same way.  With a query of matching signs the rounding errors add up instead of cancelling:
    |error| = sum |delta_i q_i| -- the situation F32_ERR_F16X2 = 2^-11 (+ accumulation) is the bound for."""
    rows = np.zeros((n_rows, 256), dtype=np.float64)
    for r in range(n_rows):
        i = int(rng.integers(0, 4))
        j = 3 * (2 * i + 1) + (0 if r % 2 else -1)          # (2j + 1) = 6 (2i + 1) -+ 1: the norm is off by 2^-14 only
        a = 2.0 ** -3 * (1 + (2 * i + 1) * 2.0 ** -11)       # 48 elements just above 2^-3: midpoint of the fp16 grid after x 2^10
        b = 2.0 ** -4 * (1 - (2 * j + 1) * 2.0 ** -12)       # 64 elements just below 2^-4
        pos = rng.permutation(256)
        sign = rng.choice([-1.0, 1.0], size=256)
        rows[r, pos[:48]] = a * sign[pos[:48]]
        rows[r, pos[48:112]] = b * sign[pos[48:112]]
    return rows


def _worst_case_rows_bf16x3(rng, n_rows):
    """Unit rows (to within f32 rounding: one free element absorbs the rest of the norm) whose elements carry the bit
    pattern that maximises what bf16 x 3 drops: x = hi + lo + r with lo ~ 2^-8 |x| (the residual just below half a bf16
    ulp, so hi rounds DOWN) and r = 0.75 * 2^-17 |x| of the same sign (lo rounds down too).  With q = x the dropped terms
    lo.lo + 2 r.x are all positive: error ~ 2^-15 |x||q| = 3.1e-5, the worst this scheme can do -- against a bound of 1.5e-4."""
    rows = np.zeros((n_rows, 256), dtype=np.float64)
    frac = 2.0 ** -8 - 2.0 ** -16 + 0.75 * 2.0 ** -17       # bits below the 7 fraction bits of hi
    for r in range(n_rows):
        while True:
            n = 62
            t = rng.integers(0, 3, size=n)                   # the top 7 fraction bits are free: they tune the norm
            mags = 2.0 ** -3 * (1 + t * 2.0 ** -7 + frac)
            rest = 1.0 - float((mags ** 2).sum())
            if 2.0 ** -8 < rest < 2.0 ** -5:
                break
        pos = rng.permutation(256)
        sign = rng.choice([-1.0, 1.0], size=256)
        rows[r, pos[:n]] = mags * sign[pos[:n]]
        rows[r, pos[n]] = np.sqrt(rest) * sign[pos[n]]       # the free element
    return rows


def test_constructed_worst_case_rows_stay_inside_the_certificate_bound(gpu_ctx, nominate_with):
    """the certificate's bounds were only ever checked on random corpora.  Here the rows are BUILT to make
    every rounding of the nominating arithmetic err in the same direction (see the two generators), at unit norm and scaled
    by powers of two and by 3.7 / 1e-3 (non-unit norms: the kernels normalise), and the queries are the rows themselves and
    sign-matched vectors of other magnitudes.  Asserted: |nominating - exact| <= the compiled-in bound in every mode, the
    construction really is adversarial in the mode it targets (several times the random-corpus maximum), and searches over
    these rows still return exactly the oracle's answer."""
    import semtools_amd as smt

    rng = np.random.default_rng(2024)
    f16x2 = getattr(gpu_ctx, "_f16x2_mode", False)
    f16x1 = getattr(gpu_ctx, "_f16x1_mode", False)
    bound = gpu_ctx._nominating_bound
    base16 = _worst_case_rows_f16x2(rng, 256)
    base_bf = _worst_case_rows_bf16x3(rng, 256)
    scales = np.array([1.0, 2.0 ** 5, 2.0 ** -7, 3.7, 1e-3])
    rows64 = np.concatenate([base16 * s for s in scales] + [base_bf * s for s in scales])
    emb = np.ascontiguousarray(rows64.astype(np.float32))
    # queries: 8 adversarial rows of each kind (cos = 1 with their own row), 8 sign-matched vectors per kind (cos ~ 0.95)
    qs = [base16[:8], base_bf[:8]]
    for base in (base16, base_bf):
        mag = np.where(base[8:16] != 0, 0.5 + rng.random((8, 256)), 0.0)
        qs.append(np.sign(base[8:16]) * mag / 16.0)
    qs = np.ascontiguousarray(np.concatenate(qs).astype(np.float32))
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    got = c.debug_batched_scores(qs).astype(np.float64)                   # [rows, 32]
    e64, q64 = emb.astype(np.float64), qs.astype(np.float64)
    cos = (e64 @ q64.T) / (np.linalg.norm(e64, axis=1)[:, None] * np.linalg.norm(q64, axis=1)[None, :])
    exact = np.maximum(1.0 - cos, 0.0)
    err = np.abs(got - exact)
    # the pairs the construction aims at: row i of a power-of-two-scaled block against "its" query
    n = 256
    own16 = max(err[blk * n + i, i] for blk in range(3) for i in range(8))                 # f16 rows x their own rows as queries
    own_bf = max(err[(5 + blk) * n + i, 8 + i] for blk in range(3) for i in range(8))
    matched16 = max(err[blk * n + 8 + i, 16 + i] for blk in range(3) for i in range(8))
    print(f"max |nominating - exact| = {err.max():.3e} (bound {bound:.1e}); f16-midpoint rows vs own query {own16:.3e}, "
          f"vs sign-matched query {matched16:.3e}; bf16-residual rows vs own query {own_bf:.3e}")
    assert err.max() <= bound, (err.max(), bound)
    if f16x2:
        assert own16 > 3.0e-4 and matched16 > 2.0e-4          # ~ 0.85 * 2^-11: adversarial indeed (random corpora: 2.3e-4 at most)
    elif f16x1:
        assert own16 > 6.0e-4                                  # the query (= the row) rounds the same way once more: ~ 2 x 0.85 x 2^-11
    elif nominate_with:
        assert own_bf > 2.0e-5                                 # ~ 2^-15 (random corpora: 1.0e-5 at most)
    # ... and the answers over these rows are still exact: indices and f64 distances of the oracle
    got_search = c.search(qs[[0, 9, 17, 25]], top_k=12)
    for (r, d), qi in zip(got_search, (0, 9, 17, 25)):
        orows, odist = _oracle_topk(emb, qs[qi], 12)
        assert r.tolist() == orows
        np.testing.assert_allclose(d, odist, rtol=0, atol=1e-12)
    c.close()


@pytest.mark.parametrize("nq", [2, 3, 5, 7])
def test_two_to_seven_queries_take_the_batched_path_on_large_shards(gpu_ctx, nq, nominate_with):
    """A K2 pass slows down with every query it carries; with the row-register kernel one K3 pass is cheaper from
    3 queries on 1 M rows, from 2 on 4 M (topk_dispatch, tuning keys gemm_min_nq / gemm_min_rows_small).  Forced
    here on a small corpus: same answers, and the batched kernels really ran."""
    import semtools_amd as smt

    emb = synth.unit_rows(30000, seed=41)
    qs = synth.unit_query(90 + nq, nq=nq)
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    via_k2 = c.search(qs, top_k=7)
    gpu_ctx.set_tuning("gemm_min_rows_small", 0)
    gpu_ctx.set_tuning("gemm_min_nq", 2)
    try:
        gpu_ctx.prof_enable(True)
        gpu_ctx.prof_reset()
        via_k3 = c.search(qs, top_k=7)
        launches, _ = gpu_ctx.prof_read("gemm")
        gpu_ctx.prof_enable(False)
    finally:
        gpu_ctx.set_tuning("gemm_min_rows_small", 1_000_000)
        gpu_ctx.set_tuning("gemm_min_nq", 5)
    if nominate_with and gpu_ctx_is_rowreg(gpu_ctx):
        assert launches > 0            # really went through K3
    for a, b in zip(via_k2, via_k3):
        assert a[0].tolist() == b[0].tolist() and np.array_equal(a[1], b[1])
    c.close()


def gpu_ctx_is_rowreg(ctx):
    return getattr(ctx, "_rowreg_mode", True)


def test_random_shapes_against_the_oracle(gpu_ctx, nominate_with):
    """Seeded random (rows, queries, k) triples -- ragged tile counts, single-tile corpora, batches that stream query
    tiles, planted duplicate / zero rows -- all through K3 (forced from 2 queries on): indices and distances = oracle."""
    import semtools_amd as smt

    rng = np.random.default_rng(20260923)
    gpu_ctx.set_tuning("gemm_min_rows_small", 0)
    gpu_ctx.set_tuning("gemm_min_nq", 2)
    try:
        for case in range(10):
            n_rows = int(rng.choice([1, 31, 33, 1000, 4095, 4096, 4097, 20001, 65537])) if case < 9 else 150000
            nq = int(rng.choice([2, 3, 9, 31, 32, 33, 64, 129, 161, 260]))
            k = int(rng.integers(1, min(56, n_rows) + 1))
            emb = synth.unit_rows(n_rows, seed=1000 + case, dup_frac=0.02, zero_frac=0.002)
            qs = synth.unit_query(2000 + case, nq=nq)
            if case % 3 == 0:
                qs[0] = 0.0                                        # a zero query rides along
            c = smt.Corpus(gpu_ctx)
            c.append(emb)
            got = c.search(qs, top_k=k)
            for i in sorted(set([0, 1, nq // 2, nq - 1])):
                orows, odist = _oracle_topk(emb, qs[i], k)
                assert got[i][0].tolist() == orows, (case, n_rows, nq, k, i)
                assert np.array_equal(got[i][1], np.array(odist)), (case, n_rows, nq, k, i)
            c.close()
    finally:
        gpu_ctx.set_tuning("gemm_min_rows_small", 1_000_000)
        gpu_ctx.set_tuning("gemm_min_nq", 5)


@pytest.mark.parametrize("stride", [2, 16])
def test_device_form_counts_an_overflowed_query_instead_of_synchronising(gpu_ctx, stride):
    """The batched device form (smt_search_topk_device: nothing synchronises) on the corpus that floods the candidate buffers: until
    round 5 launch_gemm_topk read an overflow flag back after every batch (one host synchronisation per call) and re-answered such
    queries with the scan kernel; now the final select flags them like a failed certificate -- counted by the context for the device
    form, re-answered exhaustively by the host form (the test above).  Every query is either exact or counted."""
    import torch
    import semtools_amd as smt

    try:
        n, qs, emb = _flooding_corpus(gpu_ctx, stride)
        c = smt.Corpus(gpu_ctx)
        c.append(emb)
        qd = torch.from_numpy(qs).cuda()
        o_r = torch.zeros((9, 10), dtype=torch.int64, device="cuda")
        o_d = torch.zeros((9, 10), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        gpu_ctx.uncertain_count()
        st = torch.full((9,), 7, dtype=torch.int32, device="cuda")
        c.search_topk_device(qd.data_ptr(), 9, 10, 0, o_r.data_ptr(), o_d.data_ptr(), out_status_ptr=st.data_ptr())
        gpu_ctx.synchronize()
        flagged = gpu_ctx.uncertain_count()
        inexact = 0
        rows = o_r.cpu().numpy()
        status = st.cpu().tolist()
        for i in range(9):
            res = orc.search_documents(emb, [n], qs[i], 0, 10, accurate=True)
            wrong = rows[i].tolist() != [r["match_line"] for r in res]
            inexact += wrong
            assert not (wrong and status[i] == 0), i               # per query (_ex): an answer that says "proved" IS exact
        assert flagged >= max(inexact, 1), (flagged, inexact)      # this corpus does overflow: at least one query is counted
        assert set(status) <= {0, 1, 2} and 2 in status and sum(x != 0 for x in status) == flagged, status   # ... as SMT_STATUS_OVERFLOW
        c.close()
    finally:
        gpu_ctx.set_tuning("gemm_boot_fine", 1)
