"""Range-filtered BATCHES (the workspace path-subset filter with several queries: src/workspace/store.rs:507-515 via
src/search/mod.rs:211-213) on the row-register MFMA kernel: the kernel walks a tile table (aligned 32-row tile | mask of the wanted
rows) instead of every tile, so filtered batches get the fp16 nomination modes and the corpus' operand image.  Bar: rows exactly the
oracle's on the eligible rows, f64 distances bit-equal; identical with and without the image, with the round-2 LDS-row route
(tuning key gemm_rowreg = 0) and with the single-query scan path."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import synth

pytestmark = pytest.mark.gpu


def _oracle_topk(emb, q, k):
    res = orc.search_documents(emb, [len(emb)], q, n_lines=0, top_k=k, accurate=True)
    return [r["match_line"] for r in res], [r["distance"] for r in res]


def _docs(n_rows, sizes_seed, lo, hi):
    """Cut [0, n_rows) into consecutive "documents" of lo..hi rows."""
    rng = np.random.default_rng(sizes_seed)
    out, b = [], 0
    while b < n_rows:
        e = min(n_rows, b + int(rng.integers(lo, hi + 1)))
        out.append((b, e))
        b = e
    return out


def _kernels_of(ctx, fn):
    """(result of fn(), {"gemm": launches of an MFMA kernel, "scan": launches of the scan kernel}) -- WHICH path answered: a batch
    the MFMA path refuses falls back to the scan kernel and would pass every parity check."""
    ctx.set_tuning("prof_every", 1)
    ctx.prof_enable(True)
    ctx.prof_reset()
    try:
        out = fn()
        return out, {name: ctx.prof_read(name)[0] for name in ("gemm", "scan")}
    finally:
        ctx.prof_enable(False)


def _check(c, emb, qs, ranges, k, which=None, expect_mfma=True, allow_scan=False):
    elig = np.concatenate([np.arange(b, e) for b, e in ranges])
    got, ran = _kernels_of(c.ctx, lambda: c.search(qs, top_k=k, ranges=ranges))
    if expect_mfma and len(qs) >= 8:
        assert ran["gemm"] > 0 and (allow_scan or ran["scan"] == 0), ran
    for i in (range(len(qs)) if which is None else which):
        orows, odist = _oracle_topk(emb[elig], qs[i], k)
        assert got[i][0].tolist() == elig[np.array(orows, dtype=np.int64)].tolist(), (k, i)
        assert np.array_equal(got[i][1], np.array(odist)), (k, i)
    return got


@pytest.mark.parametrize("nq", [8, 40, 130, 300])
@pytest.mark.parametrize("image", [False, True])
def test_subset_of_small_documents_on_the_tile_table(gpu_ctx, nq, image):
    """Documents of 3..60 lines, every second one wanted: almost every tile holds wanted and unwanted rows, ranges meet inside
    tiles, the last document ends in a ragged tile.  8 / 40 queries nominate with bf16 x 3 (f16 x 2 with the image), 130 with
    f16 x 2 (f16 x 1 with the image), 300 with f16 x 1."""
    import semtools_amd as smt

    n = 60_013
    emb = synth.unit_rows(n, seed=31)
    qs = synth.unit_query(32, nq=nq)
    docs = _docs(n, 5, 3, 60)
    ranges = docs[::2]
    if ranges[-1] != docs[-1]:
        ranges.append(docs[-1])            # the ragged end of the corpus is wanted
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    if image:
        c.prepack()
        assert c.image_bytes > 0
    try:
        for k in (1, 10):
            got = _check(c, emb, qs, ranges, k, which=range(0, nq, max(1, nq // 12)))
            # the single-query scan path over the same ranges
            one = c.search(qs[1], top_k=k, ranges=ranges)[0]
            assert got[1][0].tolist() == one[0].tolist() and np.array_equal(got[1][1], one[1])
            # the LDS-row route over the chunk table (what filtered batches took before): identical lists
            gpu_ctx.set_tuning("gemm_rowreg", 0)
            try:
                old = c.search(qs[:64], top_k=k, ranges=ranges)
            finally:
                gpu_ctx.set_tuning("gemm_rowreg", 1)
            for a, b in zip(got[:64], old):
                assert a[0].tolist() == b[0].tolist() and np.array_equal(a[1], b[1])
    finally:
        c.close()


def test_tile_table_corner_cases(gpu_ctx):
    """One-row ranges, ranges that share a tile with both neighbours, a range inside the previous range's last tile, ranges that
    start / end exactly on tile borders, one range covering everything, a single row wanted in the whole corpus."""
    import semtools_amd as smt

    n = 4_100
    emb = synth.unit_rows(n, seed=41)
    emb[64] = 0.0                       # a zero row inside a wanted range
    qs = synth.unit_query(42, nq=12)
    qs[5] = 0.0                         # a zero query
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    cases = [
        [(0, n)],
        [(0, 32), (32, 64), (64, 96)],
        [(5, 6), (7, 8), (9, 10), (31, 33), (33, 34), (63, 64), (64, 65), (100, 400)],
        [(10, 40), (41, 42), (43, 50), (50, 63), (63, 64), (200, 1000), (1000, 1001), (4095, 4100)],
        [(4099, 4100)],
        [(31, 32)],
        [(0, 1), (4096, 4100)],
        [(1, 3000), (3001, 3002), (3003, 3004), (3010, 3040), (3050, 4000)],
    ]
    try:
        for image in (False, True):
            if image:
                c.prepack()
            for ranges in cases:
                n_el = sum(e - b for b, e in ranges)
                for k in (1, 7):
                    # (the zero query ties with every row: past 2048 rows its candidate list overflows and the scan kernel re-answers it)
                    _check(c, emb, qs, ranges, min(k, 56), which=None, allow_scan=True)
                assert n_el > 0
    finally:
        c.close()


def test_sparse_subsets_keep_the_chunk_route_and_agree(gpu_ctx):
    """Every tenth 3-line document: the wanted rows fill < 1/4 of the tiles they touch -- the LDS-row kernel gathers 4-row chunks
    instead (common.h tiles_dense).  Same contract."""
    import semtools_amd as smt

    n = 40_000
    emb = synth.unit_rows(n, seed=51)
    qs = synth.unit_query(52, nq=20)
    ranges = [(b, b + 3) for b in range(0, n - 3, 30)]
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    try:
        _check(c, emb, qs, ranges, 5)             # (an MFMA kernel either way: the LDS-row one here)
    finally:
        c.close()


@pytest.mark.parametrize("image", [False, True])
def test_workspace_mode_over_a_subset_batched(gpu_ctx, image):
    """Store::search_line_embeddings semantics (score threshold, then top-k always: store.rs:500-523, 543) for a batch over a
    document subset: every query equals its own single-query call and the oracle's store search."""
    import semtools_amd as smt

    n = 50_000
    emb = synth.unit_rows(n, seed=61, dup_frac=0, zero_frac=0)   # (the numpy restatement below has no tie / zero-row rules)
    qs = synth.unit_query(62, nq=24)
    docs = _docs(n, 7, 20, 400)
    ranges = docs[1::2]
    elig = np.concatenate([np.arange(b, e) for b, e in ranges])
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    if image:
        c.prepack()
    try:
        for max_d in (0.80, 0.95):
            got, ran = _kernels_of(gpu_ctx, lambda: c.search(qs, top_k=6, max_distance=max_d, mode=smt.MODE_WORKSPACE, ranges=ranges))
            assert ran["gemm"] > 0 and ran["scan"] == 0, ran
            for i in range(len(qs)):
                one = c.search(qs[i], top_k=6, max_distance=max_d, mode=smt.MODE_WORKSPACE, ranges=ranges)[0]
                assert got[i][0].tolist() == one[0].tolist() and np.array_equal(got[i][1], one[1]), (max_d, i)
                # independent f64 restatement: score = 1 - d (f32 compare as qdrant does), keep score > 1 - max_d, best 6
                d = 1.0 - (emb[elig].astype(np.float64) @ qs[i].astype(np.float64)) / (
                    np.linalg.norm(emb[elig].astype(np.float64), axis=1) * np.linalg.norm(qs[i].astype(np.float64)))
                order = np.lexsort((elig, d))
                keep = [j for j in order[:64] if (1.0 - d[j]) > float(np.float32(1.0) - np.float32(max_d))][:6]
                assert got[i][0].tolist() == elig[keep].tolist(), (max_d, i)
    finally:
        c.close()


@pytest.mark.parametrize("subset", [False, True])
def test_a_level_run_in_two_parts_answers_like_the_whole_level(gpu_ctx, subset):
    """A bootstrap plan runs its first appended level in two parts with a select pass after the first quarter (tuning key
    gemm_split_last = 2) once the level holds >= 12 tiles per wave and the batch more query tiles than ring slots.  300 queries over
    1.7 M rows -- all of them, or every second 1000-row document through the tile table: the split plan launches more MFMA kernels
    than the whole-level plan (= 1), both give identical rows and distances, and the rows of a few queries are those of an fp64
    top-k over the eligible rows on the device."""
    import torch

    import semtools_amd as smt

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(21)
    rows = 1_700_000 if not subset else 3_400_000
    x = torch.randn(rows, 256, device=dev, generator=g)
    x /= x.norm(dim=1, keepdim=True)
    q = torch.randn(300, 256, device=dev, generator=g)
    q /= q.norm(dim=1, keepdim=True)
    torch.cuda.synchronize()
    ranges = smt.PackedRanges([(d * 1000, (d + 1) * 1000) for d in range(0, rows // 1000, 2)]) if subset else None
    qh = q.cpu().numpy()
    c = smt.Corpus(gpu_ctx, device_ptr=x.data_ptr(), rows=rows)
    c.prepack()
    try:
        out, launches = {}, {}
        for v in (1, 2):
            gpu_ctx.set_tuning("gemm_split_last", v)
            out[v], ran = _kernels_of(gpu_ctx, lambda: c.search(qh, top_k=10, ranges=ranges))
            assert ran["gemm"] > 0 and ran["scan"] == 0, ran
            launches[v] = ran["gemm"]
        assert launches[2] == launches[1] + 1, launches          # the level really ran in two parts
        for i in range(len(qh)):
            assert out[1][i][0].tolist() == out[2][i][0].tolist() and np.array_equal(out[1][i][1], out[2][i][1]), i
        elig = None
        if subset:
            elig = torch.zeros(rows, dtype=torch.bool, device=dev)
            elig.view(-1, 1000)[0::2] = True
        for i in (0, 151, 299):
            ref = 1.0 - (x.double() @ q[i].double())
            if elig is not None:
                ref[~elig] = 9.0
            tv, ti = torch.topk(ref, 10, largest=False)
            assert out[2][i][0].astype(np.int64).tolist() == ti.cpu().tolist(), i
            assert np.abs(out[2][i][1] - tv.cpu().numpy()).max() < 1e-6, i
    finally:
        gpu_ctx.set_tuning("gemm_split_last", 2)
        c.close()


def test_a_range_list_seen_twice_is_kept_on_the_device_and_answers_the_same(gpu_ctx):
    """The path subset of a workspace session repeats (src/workspace/store.rs:495 chunks the same paths every call): the second time a
    corpus sees a range list it keeps the list, its prefixes and its tile / chunk tables on the device (common.h RangeSet); from the
    third search on only the queries go up.  Same answers on every sight, for one query (scan kernel, chunk table) and for batches
    (MFMA kernel, tile table), with and without the image; at most four lists are kept, the least recently used one goes; a list
    that reaches past a truncated corpus is refused even though it is kept."""
    import semtools_amd as smt

    n = 60_000
    emb = synth.unit_rows(n, seed=51)
    qs = synth.unit_query(52, nq=40)
    docs = _docs(n, 53, 20, 400)
    lists = [docs[i::5 + i] for i in range(6)]                    # six different subsets
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    try:
        assert c.range_sets() == (0, 0, 0)
        first = _check(c, emb, qs, lists[0], 10, which=[0, 7, 39])
        assert c.range_sets() == (0, 0, 0)                        # seen once: nothing kept
        second = _check(c, emb, qs, lists[0], 10, which=[0, 7, 39])
        assert c.range_sets() == (1, 0, 1)                        # built on the second sight
        third, ran = _kernels_of(c.ctx, lambda: c.search(qs, top_k=10, ranges=lists[0]))
        assert c.range_sets() == (1, 1, 1) and ran["gemm"] > 0 and ran["scan"] == 0
        for a, b, d in zip(first, second, third):
            assert a[0].tolist() == b[0].tolist() == d[0].tolist() and np.array_equal(a[1], b[1]) and np.array_equal(a[1], d[1])
        # one query over the kept list: the scan kernel and the set's CHUNK table (built on this first use), workspace mode too
        for _ in range(2):
            _check(c, emb, qs[:1], lists[0], 7, expect_mfma=False)
        ws = c.search(qs[:3], top_k=5, max_distance=0.9, mode=1, ranges=lists[0])
        assert c.range_sets()[1] >= 4
        c.prepack()                                               # the same kept tile table under the operand image
        img = _check(c, emb, qs, lists[0], 10, which=[0, 7, 39])
        for a, b in zip(first, img):
            assert a[0].tolist() == b[0].tolist() and np.array_equal(a[1], b[1])
        ws_img = c.search(qs[:3], top_k=5, max_distance=0.9, mode=1, ranges=lists[0])
        for a, b in zip(ws, ws_img):
            assert a[0].tolist() == b[0].tolist() and np.array_equal(a[1], b[1])
        # five more lists, each searched three times: four sets stay, list 0 (least recently used) has gone and comes back
        for lst in lists[1:]:
            for _ in range(3):
                _check(c, emb, qs[:9], lst, 6, which=[0, 8])
        kept, hits, builds = c.range_sets()
        assert kept == 4 and builds == 6
        _check(c, emb, qs[:9], lists[0], 6, which=[0, 8])
        _check(c, emb, qs[:9], lists[0], 6, which=[0, 8])
        assert c.range_sets()[2] == builds + 1 and c.range_sets()[0] == 4
        # a subset that grows by one document is a different list
        grown = sorted(lists[0] + [d for d in docs if d not in lists[0]][:1])
        _check(c, emb, qs[:9], grown, 6, which=[0, 8])
        # truncation below a kept list's last range: refused
        last_end = max(e for _, e in lists[0])
        c.truncate(last_end - 1)
        with pytest.raises(RuntimeError):
            c.search(qs[:9], top_k=6, ranges=lists[0])
    finally:
        c.close()


def test_kept_range_sets_on_a_sharded_corpus(gpu_ctx):
    """The same on a corpus row-sharded over three logical ranks: every shard keeps ITS localised range list (the global ranges cut
    along the shard's pieces), the answers stay those of the unsharded corpus call after call."""
    import semtools_amd as smt

    n = 30_000
    emb = synth.unit_rows(n, seed=81)
    qs = synth.unit_query(82, nq=20)
    docs = _docs(n, 83, 30, 500)
    subset = docs[::2]
    plain = smt.Corpus(gpu_ctx)
    plain.append(emb)
    g = smt.Group.logical(0, 3)
    sc = smt.ShardedCorpus(g, rows=emb)
    try:
        want = plain.search(qs, top_k=9, ranges=subset)
        want1 = plain.search(qs[:1], top_k=9, ranges=subset)
        for _ in range(4):
            got = sc.search(qs, top_k=9, ranges=subset)
            got1 = sc.search(qs[:1], top_k=9, ranges=subset)
            for a, b in zip(got + got1, want + want1):
                assert a[0].tolist() == b[0].tolist() and np.array_equal(a[1], b[1])
        kept = [sc.shard(i, want_base=False)[0].range_sets() for i in range(3)]
        assert all(k[0] >= 1 and k[1] >= 2 for k in kept), kept       # every shard built a set and answered from it
    finally:
        sc.close(); g.close(); plain.close()
