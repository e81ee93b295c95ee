"""GPU parity: K1 embed (gather + mean-pool + L2-normalise) vs the oracle's pool_ids (A3/A4).
The kernel reproduces the CPU code's serial f32 chains, so the bar here is bit-exactness."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(gpu_ctx):
    import semtools_amd as smt

    table = synth.table(5000, seed=2)
    m = smt.Model(gpu_ctx, table, normalize=True)
    yield table, m
    m.close()


def test_embed_runs_of_lines_per_group_bit_exact(model):
    """More lines than the chip has lane groups: every group then walks a RUN of consecutive lines on its own (the four
    groups of a wave at different lines, finishing at different steps).  300 k ragged lines incl. empty ones, lines longer
    than the cap and a stretch of empty lines: bit-exact, with and without truncation."""
    table, m = model
    rng = np.random.default_rng(11)
    n = 300_000
    lens = rng.integers(0, 33, size=n)
    lens[rng.integers(0, n, size=200)] = rng.integers(100, 700, size=200)      # long lines scattered about
    lens[1000:1040] = 0                                                         # a stretch of empty lines
    lens[-1] = 0
    offsets = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    ids = rng.integers(0, 5000, size=int(offsets[-1])).astype(np.uint32)
    for cap in (2048, 16):
        got, _ = m.embed(ids, offsets, max_tokens=cap)
        ref = orc.embed_lines(table, ids, offsets, normalize=True, max_tokens=cap)
        assert np.array_equal(got, ref), f"cap {cap}: max |diff| {np.abs(got - ref).max()}"
        assert not got[lens == 0].any()


@pytest.mark.parametrize("shape", ["giants_among_short", "all_empty", "one_line_holds_everything", "empty_then_full"])
def test_embed_runs_of_equal_work(model, gpu_ctx, shape):
    """Runs are cut by work (tokens + 4 per line; embed_runs_kernel), not by line count: shapes whose cut points are extreme --
    2048-token lines among four-token ones (groups that own a single line beside groups that own hundreds), nothing but empty
    lines, one line holding every token (most groups find nothing to do), 30 k empty lines in front of the tokens.  Bit-exact
    against the oracle, identical to the cut by line count (tuning bit 3), with and without truncation."""
    table, m = model
    rng = np.random.default_rng(23)
    n = 40_000                                             # > 256 CUs x 64 groups: more than one line per group
    if shape == "giants_among_short":
        lens = np.full(n, 4)
        lens[rng.choice(n, size=60, replace=False)] = 2048
    elif shape == "all_empty":
        lens = np.zeros(n, dtype=np.int64)
    elif shape == "one_line_holds_everything":
        lens = np.zeros(n, dtype=np.int64)
        lens[n // 3] = 300_000
    else:
        lens = np.concatenate([np.zeros(30_000, dtype=np.int64), rng.integers(1, 40, size=n - 30_000)])
    offsets = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    ids = rng.integers(0, 5000, size=int(offsets[-1])).astype(np.uint32)
    for cap in (2048, 7):
        ref = orc.embed_lines(table, ids, offsets, normalize=True, max_tokens=cap)
        got, _ = m.embed(ids, offsets, max_tokens=cap)
        assert np.array_equal(got, ref), (shape, cap)
        gpu_ctx.set_tuning("embed_batched", 3 | 8)
        try:
            by_lines, _ = m.embed(ids, offsets, max_tokens=cap)
        finally:
            gpu_ctx.set_tuning("embed_batched", 3)
        assert np.array_equal(by_lines, ref), (shape, cap)


def test_embed_runs_too_long_for_32_bit_positions_take_the_generic_kernel(model, gpu_ctx):
    """The default kernel counts token positions in 32 bits from the first token of a group's run and leaves runs that do not fit
    (2^32 tokens) to the generic kernel launched behind it.  Tuning bit 2 of embed_batched lowers that limit to 64 tokens so the
    split is exercised at test sizes: runs of ~19 ragged lines (mostly over the limit, a stretch of empty lines under it), and one
    line per group (0..40 tokens with a few long ones): both kernels write their share, bit-exact, no row written twice wrongly."""
    table, m = model
    rng = np.random.default_rng(12)
    gpu_ctx.set_tuning("embed_batched", 7)
    try:
        for n in (300_000, 1500):
            lens = rng.integers(0, 41, size=n)
            lens[rng.integers(0, n, size=20)] = rng.integers(64, 900, size=20)
            lens[n // 3: n // 3 + 400] = 0
            offsets = np.zeros(n + 1, dtype=np.uint64)
            np.cumsum(lens, out=offsets[1:])
            ids = rng.integers(0, 5000, size=int(offsets[-1])).astype(np.uint32)
            for cap in (2048, 16):
                got, _ = m.embed(ids, offsets, max_tokens=cap)
                ref = orc.embed_lines(table, ids, offsets, normalize=True, max_tokens=cap)
                assert np.array_equal(got, ref), f"n {n} cap {cap}: max |diff| {np.abs(got - ref).max()}"
    finally:
        gpu_ctx.set_tuning("embed_batched", 3)


def test_embed_bit_exact(model):
    table, m = model
    ids, offsets = synth.token_lines(1003, V=5000, seed=1, min_tok=0, max_tok=40)
    got, _ = m.embed(ids, offsets, max_tokens=2048)
    ref = orc.embed_lines(table, ids, offsets, normalize=True, max_tokens=2048)
    assert np.array_equal(got, ref), f"max |diff| {np.abs(got - ref).max()}"
    # empty lines -> zero vectors
    empty = np.nonzero(np.diff(offsets.astype(np.int64)) == 0)[0]
    assert empty.size > 0 and not got[empty].any()


def test_embed_truncation_and_long_lines(model):
    table, m = model
    rng = np.random.default_rng(5)
    lens = np.array([0, 1, 2, 511, 512, 513, 3000, 7], dtype=np.int64)
    ids = rng.integers(0, 5000, size=int(lens.sum())).astype(np.uint32)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    for cap in (512, 2048, 0):
        got, _ = m.embed(ids, offsets, max_tokens=cap)
        ref = orc.embed_lines(table, ids, offsets, normalize=True, max_tokens=cap)
        assert np.array_equal(got, ref), cap


def test_embed_no_normalize(gpu_ctx):
    import semtools_amd as smt

    table = synth.table(300, seed=12)
    m = smt.Model(gpu_ctx, table, normalize=False)
    ids, offsets = synth.token_lines(77, V=300, seed=3)
    got, _ = m.embed(ids, offsets)
    assert np.array_equal(got, orc.embed_lines(table, ids, offsets, normalize=False))
    m.close()


def test_embed_appends_to_corpus_and_search(model, gpu_ctx):
    """create_document_from_content -> search_documents end to end on ids (mod.rs:49-120)."""
    import semtools_amd as smt

    table, m = model
    ids, offsets = synth.token_lines(500, V=5000, seed=9, min_tok=1, max_tok=30)
    c = smt.Corpus(gpu_ctx)
    _, first = m.embed(ids, offsets, append_to=c, want_host=False)
    assert first == 0 and c.rows == 500
    _, first2 = m.embed(ids[: int(offsets[10])], offsets[:11], append_to=c, want_host=False)
    assert first2 == 500 and c.rows == 510
    ref = orc.embed_lines(table, ids, offsets)
    assert np.array_equal(c.read_rows(0, 500), ref)
    # query = line 17's tokens, 512-token cap like encode_single
    q, _ = m.embed(ids[int(offsets[17]): int(offsets[18])], np.array([0, offsets[18] - offsets[17]], np.uint64), 512)
    rows, dist = c.search(q[0], top_k=3)[0]
    res = orc.search_documents(np.concatenate([ref, ref[:10]]), [510], q[0], 0, 3, accurate=True)
    assert rows.tolist() == [r["match_line"] for r in res]
    assert rows[0] in (17, 507) and dist[0] < 1e-6
    c.close()
