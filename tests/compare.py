"""Tie-aware parity comparator (north_star's parity bar).

The reference's own ordering is the stable sort of serial-f32 simsimd distances (src/search/mod.rs:86,107-111).
simsimd's SIMD backends are not bit-reproducible across CPUs, which is why the bar is: indices bit-exact EXCEPT
that rows whose reference distances differ by < 1e-5 may permute within their group; exact-duplicate rows come
back in ascending row order; distances within 1e-5.  TEST INFRASTRUCTURE."""
import numpy as np

from oracle import oracle as orc

EPS = 1e-5


def reference_distances(emb, q, accurate=False):
    """The oracle's distance of EVERY row to q (serial-f32 restatement by default), as float64[N]."""
    emb = np.ascontiguousarray(emb, dtype=np.float32)
    n = len(emb)
    res = orc.search_documents(emb, [n], q, n_lines=0, top_k=max(n, 1), accurate=accurate)
    out = np.full(n, np.nan)
    for r in res:
        out[r["match_line"]] = r["distance"]
    assert not np.isnan(out).any()
    return out


def assert_topk_tie_aware(got_rows, got_dist, ref_all, k, eps=EPS, rows_subset=None):
    """got_rows/got_dist: what the library returned for top-k.  ref_all[r]: the reference distance of row r.
    rows_subset: the rows that were eligible (range-filtered searches); default all.
    Returns the number of positions where the row differs from the reference's own stable order (permutations
    inside tie groups -- 0 on data without near-ties)."""
    got_rows = np.asarray(got_rows, dtype=np.int64)
    got_dist = np.asarray(got_dist, dtype=np.float64)
    elig = np.arange(len(ref_all)) if rows_subset is None else np.asarray(sorted(rows_subset), dtype=np.int64)
    order = elig[np.lexsort((elig, ref_all[elig]))]          # the reference's order: distance asc, row asc (stable sort)
    ref_sorted = ref_all[order]
    n = min(int(k), len(elig))
    assert len(got_rows) == n, f"expected {n} hits, got {len(got_rows)}"
    assert len(set(got_rows.tolist())) == n, "a row was returned twice"
    assert set(got_rows.tolist()) <= set(elig.tolist()), "a row outside the eligible set was returned"
    # (1) position i holds a row whose reference distance is within eps of the reference's i-th distance
    slack = np.abs(ref_all[got_rows] - ref_sorted[:n])
    assert (slack < eps).all(), f"row at position {int(slack.argmax())} is not in the reference's tie group there ({slack.max():.3g})"
    # (2) every row that is clearly (by more than eps) inside the reference's top-k is present
    if n:
        must = set(order[:n][ref_sorted[:n] < ref_sorted[n - 1] - eps].tolist())
        missing = must - set(got_rows.tolist())
        assert not missing, f"rows {sorted(missing)[:5]} belong to the top-{n} by more than {eps} and are missing"
    # (3) exact duplicates (bit-equal returned distances) come back in ascending row order
    same = got_dist[1:] == got_dist[:-1]
    assert (got_rows[1:][same] > got_rows[:-1][same]).all(), "equal distances must be ordered by row (stable sort)"
    # (4) returned distances are sorted and within eps of the reference's
    assert (np.diff(got_dist) >= 0).all()
    assert (np.abs(got_dist - ref_all[got_rows]) < eps).all()
    return int((got_rows != order[:n]).sum())
