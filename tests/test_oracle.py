"""CPU: the oracle against (i) published FNV-1a vectors, (ii) its independent numpy twin,
(iii) the committed golden fixtures, (iv) the reference's own in-tree test cases restated
(src/search/mod.rs:218-464, src/workspace/store.rs:717-1375) on a synthetic model.

The reference's tests pin behaviour, not numbers ("parity unpinned" for the arithmetic --
oracle/semtools_oracle.h); every behavioural assertion they make is reproduced here."""
import os

import numpy as np
import pytest

from oracle import oracle as orc, oracle_np as onp
from tests import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ------------------------------------------------------------------ published vectors
def test_fnv1a_published_vectors():
    # FNV-1a 64 test vectors from the FNV reference distribution (Noll)
    assert orc.fnv1a_hash(b"") == 0xCBF29CE484222325
    assert orc.fnv1a_hash(b"a") == 0xAF63DC4C8601EC8C
    assert orc.fnv1a_hash(b"foobar") == 0x85944171F73967E8
    assert onp.fnv1a_hash(b"foobar") == 0x85944171F73967E8


def test_point_ids_follow_store_rs():
    # LineEmbedding::id = fnv1a(path bytes || i32 LE) (store.rs:82-89); DocMeta::id = fnv1a(path) (:75-80)
    for path, line in (("/test/doc1.txt", 0), ("a", 7), ("päth/ü.txt", 123456), ("x", -1)):
        want = onp.fnv1a_hash(path.encode() + int(line).to_bytes(4, "little", signed=True))
        assert orc.line_embedding_id(path, line) == want == onp.line_embedding_id(path, line)
        assert orc.doc_meta_id(path) == onp.fnv1a_hash(path.encode())
    # same path+line -> same id (upsert replaces, store.rs:951-1000); different line -> different id
    assert orc.line_embedding_id("/t/doc.txt", 3) == orc.line_embedding_id("/t/doc.txt", 3)
    assert orc.line_embedding_id("/t/doc.txt", 3) != orc.line_embedding_id("/t/doc.txt", 4)


# ------------------------------------------------------------------ C oracle vs numpy twin
def test_pool_twin_bit_exact():
    table = synth.table(300, seed=5)
    rng = np.random.default_rng(0)
    for n in (0, 1, 2, 7, 64, 513):
        ids = rng.integers(0, 300, n)
        for norm in (True, False):
            for cap in (0, 5, 512):
                a = orc.pool_ids(table, ids, norm, cap)
                b = onp.pool_ids(table, ids, norm, cap)
                assert np.array_equal(a, b), (n, norm, cap)
    assert not orc.pool_ids(table, [], True).any()  # empty line -> zero vector


def test_pool_is_unit_norm_and_order_sensitive():
    table = synth.table(300, seed=5)
    v = orc.pool_ids(table, [1, 2, 3, 4, 5], True)
    assert abs(float(np.linalg.norm(v.astype(np.float64))) - 1.0) < 1e-6
    # truncation happens BEFORE pooling (encode_with_args max_length)
    assert np.array_equal(orc.pool_ids(table, [1, 2, 3, 4, 5], True, 2), orc.pool_ids(table, [1, 2], True))


def test_cosine_twin_and_rules():
    rng = np.random.default_rng(1)
    for _ in range(50):
        a = rng.standard_normal(256).astype(np.float32)
        b = rng.standard_normal(256).astype(np.float32)
        assert abs(orc.cosine(a, b) - onp.cosine_serial(a, b)) <= 1e-7
        assert orc.cosine(a, b, True) == onp.cosine_accurate(a, b)
        assert abs(orc.cosine(a, b) - orc.cosine(a, b, True)) <= 1e-5   # the contract's tolerance
    z = np.zeros(256, np.float32)
    a = rng.standard_normal(256).astype(np.float32)
    assert orc.cosine(z, z) == 0.0 and orc.cosine(z, z, True) == 0.0      # a2 == 0 && b2 == 0 -> 0
    assert orc.cosine(a, z) == 1.0 and orc.cosine(z, a) == 1.0            # ab == 0 -> 1
    assert orc.cosine(a, a) <= 1e-6 and orc.cosine(a, a) >= 0.0           # clipped at 0
    assert 1.99 < orc.cosine(a, -a) <= 2.0


# ------------------------------------------------------------------ golden fixtures
def test_golden_embed():
    g = np.load(os.path.join(GOLD, "embed_small.npz"))
    table = synth.table(int(g["V"]), seed=int(g["table_seed"]))
    assert np.array_equal(orc.embed_lines(table, g["ids"], g["offsets"], True, 2048), g["emb"])
    assert np.array_equal(orc.embed_lines(table, g["ids"], g["offsets"], True, 4), g["emb_cap4"])


def test_golden_search():
    g = np.load(os.path.join(GOLD, "search_small.npz"))
    corpus, qs = g["corpus"], g["queries"]
    for qi in range(3):
        for k in (1, 3, 10):
            for acc, tag in ((False, "ser"), (True, "acc")):
                res = orc.search_documents(corpus, [len(corpus)], qs[qi], 3, k, accurate=acc)
                assert [r["match_line"] for r in res] == g[f"q{qi}_k{k}_{tag}_rows"].tolist()
                assert np.array_equal([r["distance"] for r in res], g[f"q{qi}_k{k}_{tag}_dist"])
        res = orc.search_documents(corpus, [len(corpus)], qs[qi], 3, 3, max_distance=0.9, accurate=True)
        assert [r["match_line"] for r in res] == g[f"q{qi}_thr0.9_rows"].tolist()


# ------------------------------------------------------------------ reference test cases, restated
class ToyModel:
    """Stand-in for StaticModel: whitespace words hashed into a synthetic table, then the oracle's
    pool step.  The real tokenizer/model are not available offline; the reference tests assert
    properties that do not depend on which embedding is used."""

    def __init__(self, V=4096, seed=2):
        self.table = synth.table(V, seed)
        self.V = V

    def ids(self, text):
        return [orc.fnv1a_hash(w.encode()) % self.V for w in text.split()]

    def encode_single(self, text):
        return orc.pool_ids(self.table, self.ids(text), True, 512)

    def encode(self, lines):
        return np.stack([orc.pool_ids(self.table, self.ids(t), True, 2048) for t in lines]) if lines else \
            np.zeros((0, 256), np.float32)


def create_document_from_content(model, content, ignore_case=False):
    """src/search/mod.rs:49-75 on the toy model: None for empty content, original lines kept."""
    lines = content.splitlines()
    if not lines:
        return None
    emb = model.encode([s.lower() for s in lines] if ignore_case else lines)
    return dict(lines=lines, embeddings=emb)


def search(model, docs, query, n_lines=3, top_k=3, max_distance=None):
    emb = np.concatenate([d["embeddings"] for d in docs]) if docs else np.zeros((0, 256), np.float32)
    return orc.search_documents(emb, [len(d["lines"]) for d in docs], model.encode_single(query), n_lines, top_k,
                                max_distance)


@pytest.fixture(scope="module")
def model():
    return ToyModel()


def test_search_documents_basic_sorted(model):                     # mod.rs:252-274
    docs = [create_document_from_content(model, "hello world\ngoodbye world\ntest line"),
            create_document_from_content(model, "another test\nmore content")]
    res = search(model, docs, "test query")
    assert res and all(res[i - 1]["distance"] <= res[i]["distance"] for i in range(1, len(res)))


def test_search_documents_with_max_distance_is_strict(model):      # mod.rs:276-293
    docs = [create_document_from_content(model, "line 1\nline 2\nline 3")]
    res = search(model, docs, "test", max_distance=0.5)
    assert all(r["distance"] < 0.5 for r in res)
    every = search(model, docs, "line 1", max_distance=100.0)
    assert len(every) == 3                                           # threshold mode returns ALL hits, not top_k
    d0 = every[0]["distance"]
    assert [r for r in search(model, docs, "line 1", max_distance=d0)] == []  # strict <


def test_search_documents_top_k_limit(model):                       # mod.rs:295-313
    docs = [create_document_from_content(model, "line 1\nline 2\nline 3\nline 4\nline 5")]
    assert len(search(model, docs, "test", top_k=2)) <= 2
    assert len(search(model, docs, "test", top_k=0)) == 0


def test_search_result_context_calculation(model):                  # mod.rs:315-335
    docs = [create_document_from_content(model, "\n".join(f"line {i}" for i in range(6)))]
    res = search(model, docs, "line 3", n_lines=1)
    assert res[0]["match_line"] == 3 and res[0]["end"] - res[0]["start"] == 3


def test_context_at_file_boundaries(model):                         # mod.rs:337-357
    docs = [create_document_from_content(model, "first\nsecond")]
    res = search(model, docs, "first", n_lines=5)
    assert res[0]["start"] == 0 and res[0]["end"] == 2


def test_multiple_documents_search(model):                          # mod.rs:359-379
    docs = [create_document_from_content(model, "apple fruit\nbanana"), create_document_from_content(model, "orange fruit\ngrape")]
    res = search(model, docs, "fruit", top_k=3)
    assert {r["doc"] for r in res} == {0, 1}


def test_empty_documents_handling(model):                           # mod.rs:381-391
    assert search(model, [], "test") == []


def test_case_insensitive_search(model):                            # mod.rs:393-415
    doc = create_document_from_content(model, "Hello World\nGOODBYE WORLD\nTest Line", ignore_case=True)
    res = search(model, [doc], "hello world".lower())
    assert res and res[0]["match_line"] == 0 and res[0]["distance"] < 1e-6


def test_create_document_from_content(model):                       # mod.rs:417-464
    doc = create_document_from_content(model, "Line 1\nLine 2\nLine 3")
    assert doc["lines"] == ["Line 1", "Line 2", "Line 3"] and len(doc["embeddings"]) == 3
    assert create_document_from_content(model, "") is None
    doc = create_document_from_content(model, "Hello World\nGOODBYE world", ignore_case=True)
    assert doc["lines"] == ["Hello World", "GOODBYE world"] and len(doc["embeddings"]) == 2


def test_stable_sort_keeps_document_order_for_ties(model):          # mod.rs:107-111 (stable sort_by)
    docs = [create_document_from_content(model, "same words\nother"), create_document_from_content(model, "x\nsame words")]
    res = search(model, docs, "same words", top_k=2)
    assert [(r["doc"], r["match_line"]) for r in res] == [(0, 0), (1, 1)] and res[0]["distance"] == res[1]["distance"]


def test_store_known_answer():                                      # store.rs:814-850
    emb = np.stack([np.full(256, v, np.float32) for v in (0.1, 0.5, 0.75)])
    res = orc.search_line_embeddings(emb, [0, 1, 2], [0, 1, 2], np.full(256, 0.1, np.float32), [0], 1, 0.1)
    assert len(res) == 1 and res[0]["path_id"] == 0 and res[0]["line_number"] == 0 and res[0]["distance"] < 0.1
    # all three stored vectors are colinear with the query: only the path filter discriminates
    res = orc.search_line_embeddings(emb, [0, 1, 2], [0, 1, 2], np.full(256, 0.1, np.float32), [0, 1, 2], 3, 0.1)
    assert [r["row"] for r in res] == [0, 1, 2]


def test_store_search_semantics():                                  # store.rs:489-491, :521, :543
    emb = synth.unit_rows(500, seed=1)
    q = synth.unit_query(2)[0]
    path = (np.arange(500) // 50).astype(np.uint32)
    line = (np.arange(500) % 50).astype(np.int32)
    assert orc.search_line_embeddings(emb, path, line, q, [], 3) == []
    assert orc.search_line_embeddings(emb, path, line, q, [1], 0) == []
    res = orc.search_line_embeddings(emb, path, line, q, [1, 3], 5, 0.99)
    assert len(res) <= 5 and all(r["path_id"] in (1, 3) for r in res)      # top_k applies even with a threshold
    assert all(res[i - 1]["distance"] <= res[i]["distance"] for i in range(1, len(res)))
    # > 1000 paths are chunked (store.rs:495) and still give the global top-k
    big = orc.search_line_embeddings(emb, np.arange(500, dtype=np.uint32) * 7, line, q,
                                     (np.arange(500) * 7).astype(np.uint32), 4)
    allp = orc.search_documents(emb, [500], q, 0, 4, accurate=False)
    assert [r["row"] for r in big] == [r["match_line"] for r in allp]


def test_fair_cpu_scan_agrees_with_restatement():
    emb = synth.unit_rows(5000, seed=8)
    q = synth.unit_query(9)[0]
    rows, dist = orc.scan_topk_threads(emb, q, 10, 4)
    ref = orc.search_documents(emb, [5000], q, 0, 10)
    assert rows.tolist() == [r["match_line"] for r in ref]
    assert np.allclose(dist, [r["distance"] for r in ref], rtol=0, atol=1e-6)


# ------------------------------------------------------------------ the REAL reference's outputs (oracle/_ref recipe)
# tests/golden/ref_*.npz are written by oracle/_ref/collect.py from a run of the real reference (semtools v3.0.0 +
# model2vec-rs 0.1.3 + simsimd 6.5.1 + qdrant-edge) on the committed synthetic inputs.  The round's container cannot produce
# them (no cargo, no network): until somebody runs the recipe these tests SKIP with "parity unpinned", and DESIGN.md section 6
# says the same.  Once the fixtures exist they are the anchor every other parity test inherits from.
UNPINNED = ("parity unpinned: tests/golden/{} has not been generated -- run the recipe in oracle/_ref/README.md "
            "(needs cargo + crates.io); the oracle is checked against its numpy twin and the reference's behavioural tests only")


def _ref_fixture(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        print(UNPINNED.format(name))
        pytest.skip(UNPINNED.format(name))
    return np.load(path, allow_pickle=True)


def test_reference_model2vec_embeddings_match_the_oracle_bit_for_bit():
    """model2vec-rs 0.1.3: from_pretrained + encode_with_args(.., Some(2048 / 512 / 4), ..) + encode_single on 400 lines
    (empty, all-unknown, 3000 tokens) vs orc_embed_lines on the ids the SAME tokenizer.json gives (tokenizers wheel = the crate
    model2vec-rs uses; add_special_tokens = false, unk ids dropped, truncated to the cap)."""
    import json

    ref = _ref_fixture("ref_embed.npz")
    from tokenizers import Tokenizer, models, pre_tokenizers

    V = int(ref["V"])
    table = synth.table(V, seed=int(ref["table_seed"]))
    vocab = {f"w{i}": i for i in range(V - 1)}
    vocab["[UNK]"] = V - 1
    tok = Tokenizer(models.WordLevel(vocab=vocab, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()

    def csr(texts, cap):
        ids, offsets = [], [0]
        for t in texts:
            # model2vec-rs truncates the STRING to cap * median token length characters first, then encodes, drops unk, truncates
            got = [i for i in tok.encode(t, add_special_tokens=False).ids if i != V - 1][:cap]
            ids += got
            offsets.append(len(ids))
        return np.array(ids, np.uint32), np.array(offsets, np.uint64)

    lines, queries = [str(x) for x in ref["lines"]], [str(x) for x in ref["queries"]]
    for key, cap in (("emb_2048", 2048), ("emb_512", 512), ("emb_4", 4)):
        ids, offsets = csr(lines, cap)
        mine = orc.embed_lines(table, ids, offsets, True, cap)
        assert np.array_equal(mine, ref[key]), (key, float(np.abs(mine - ref[key]).max()))
    ids, offsets = csr(queries, 512)
    assert np.array_equal(orc.embed_lines(table, ids, offsets, True, 512), ref["emb_single"])
    print("pinned against", json.loads(str(ref["versions"])))


def test_reference_simsimd_cosine_and_search_documents_match_the_oracle():
    """simsimd 6.5.1 cosine on 3 x 600 pairs (duplicates, zero rows): <= 1e-5 to both oracle forms, zero rules exact;
    semtools v3.0.0 search_documents: indices / windows exact, distances <= 1e-5."""
    import json

    ref = _ref_fixture("ref_search.npz")
    corpus, qs, cos = ref["corpus"], ref["queries"], ref["simsimd_cosine"]
    for qi in range(len(qs)):
        for r in range(len(corpus)):
            for acc in (False, True):
                assert abs(orc.cosine(qs[qi], corpus[r], accurate=acc) - cos[qi, r]) <= 1e-5, (qi, r, acc)
        zero = np.nonzero(~corpus.any(axis=1))[0]
        assert len(zero) and all(cos[qi, z] == 1.0 for z in zero)          # ab == 0 -> distance 1
    split = 400
    for case in json.loads(str(ref["search_documents"])):
        res = orc.search_documents(corpus, [split, len(corpus) - split], qs[case["query"]], n_lines=case["n_lines"],
                                   top_k=case["top_k"], max_distance=case["max_distance"], accurate=True)
        hits = case["hits"]
        assert [(f"doc{r['doc']}", r["match_line"], r["start"], r["end"]) for r in res] == \
               [(h["filename"], h["match_line"], h["start"], h["end"]) for h in hits], case
        assert np.allclose([r["distance"] for r in res], [h["distance"] for h in hits], rtol=0, atol=1e-5)


def test_reference_store_search_matches_the_oracle():
    """semtools v3.0.0 Store::search_line_embeddings (qdrant-edge) incl. the reference's own known-answer vectors."""
    import json

    ref = _ref_fixture("ref_store.npz")
    rows, qs = ref["store_rows"], ref["queries"]
    names = ["doc1", "doc2", "doc3"]
    path = np.array([0, 1, 1] + [2] * (len(rows) - 3), np.uint32)
    line = np.array([0, 0, 1] + list(range(len(rows) - 3)), np.int32)
    probes = {"q_0.1": np.full(256, 0.1, np.float32), "row10": rows[10], "vecq0": qs[0]}
    for case in json.loads(str(ref["store_search"])):
        subset = np.array([names.index(p) for p in case["subset"]], np.uint32)
        res = orc.search_line_embeddings(rows, path, line, probes[case["query"]], subset, case["top_k"], case["max_distance"])
        hits = case["hits"]
        assert [(names[r["path_id"]], r["line_number"]) for r in res] == [(h["path"], h["line_number"]) for h in hits], case
        assert np.allclose([r["distance"] for r in res], [h["distance"] for h in hits], rtol=0, atol=1e-5)


# ------------------------------------------------------------------ third implementations (neither the reference's crates nor ours)
def test_cosine_and_pool_against_scipy_and_sklearn():
    """Not a pin on the reference (its crates cannot be had here: DESIGN 6) -- a check that the oracle's two arithmetic steps agree with
    two widely used independent implementations that ARE installed: scipy's cosine distance (f64) and mean + sklearn `normalize`
    for the model2vec pool step.  Tolerances: the accurate cosine to 1e-12 of scipy on f32 inputs; the serial-f32 form within the 1e-5
    contract; the pooled vector within f32 rounding of the f64 mean / norm."""
    from scipy.spatial.distance import cosine as sp_cosine
    from sklearn.preprocessing import normalize

    rng = np.random.default_rng(31)
    for _ in range(200):
        a = rng.standard_normal(256).astype(np.float32) * float(rng.uniform(0.01, 30))
        b = rng.standard_normal(256).astype(np.float32) * float(rng.uniform(0.01, 30))
        ref = sp_cosine(a.astype(np.float64), b.astype(np.float64))
        assert abs(orc.cosine(a, b, accurate=True) - max(ref, 0.0)) <= 1e-12
        assert abs(orc.cosine(a, b, accurate=False) - max(ref, 0.0)) <= 1e-5
    table = synth.table(500, seed=7)
    for n in (1, 3, 40, 512):
        ids = rng.integers(0, 500, n)
        want = normalize(table[ids].astype(np.float64).mean(axis=0, keepdims=True))[0]
        got = orc.pool_ids(table, ids, True)
        assert np.abs(got.astype(np.float64) - want).max() <= 2e-6, n


# ------------------------------------------------------------------ outside the library's domain (semtools_amd/csrc/domain.hip)
def test_non_finite_rows_score_zero_in_every_form():
    """cos_finish's `unclipped > 0 ? unclipped : 0` (simsimd SIMSIMD_MAKE_COS, every backend's normalise step) turns a NaN into
    distance 0.0: a row with a NaN or Inf component is the reference's BEST match (src/search/mod.rs:86-89 pushes it, 0.0 < 100.0;
    :107-111 sorts it first).  Both restatements agree on that -- it is what the library REFUSES to reproduce
    (SMT_E_INVALID where such a row would enter a corpus; tests/test_gpu_domain.py)."""
    rng = np.random.default_rng(3)
    q = synth.unit_query(1)[0]
    for bad in (np.nan, np.inf, -np.inf):
        row = synth.unit_rows(1, seed=4)[0].copy()
        row[int(rng.integers(0, 256))] = bad
        for accurate in (False, True):
            assert orc.cosine(q, row, accurate=accurate) == 0.0
            assert orc.cosine(row, q, accurate=accurate) == 0.0
    emb = synth.unit_rows(50, seed=5, dup_frac=0, zero_frac=0)
    emb[37, 100] = np.nan
    res = orc.search_documents(emb, [50], q, n_lines=0, top_k=3, accurate=True)
    assert res[0]["match_line"] == 37 and res[0]["distance"] == 0.0


def test_serial_and_accurate_forms_part_company_outside_the_domain():
    """Where f32 accumulators overflow or underflow the two forms -- and simsimd's backends among themselves -- stop agreeing, so the
    1e-5 contract has nothing to hold on to: the serial-f32 form answers 1.0 (b2 = +Inf, rsqrt 0) or 0.0 (b2 underflows to 0, rsqrt +Inf, the
    clip), the f64 form the true cosine.  Inside [2^-40, 2^40] (largest magnitude of a vector) they agree to 1e-5 and a power-of-two scale
    changes neither by a bit."""
    base = synth.unit_rows(1, seed=9, dup_frac=0, zero_frac=0)[0]
    q = (base + 0.05 * synth.unit_query(2)[0]).astype(np.float32)
    true = orc.cosine(q, base, accurate=True)
    assert 0.0 < true < 0.01
    huge = (base * np.float32(1e20)).astype(np.float32)          # squares overflow f32
    assert orc.cosine(q, huge, accurate=True) == pytest.approx(true, abs=1e-7)     # (the f32 product by 1e20 rounds each component)
    assert orc.cosine(q, huge, accurate=False) == 1.0             # b2 = +Inf -> rsqrt 0 -> 1 - 0
    tiny = (base * np.float32(1e-25)).astype(np.float32)          # squares underflow to 0
    assert orc.cosine(q, tiny, accurate=True) == pytest.approx(true, abs=1e-7)
    assert orc.cosine(q, tiny, accurate=False) == 0.0             # b2 underflows to 0 -> rsqrt +Inf -> 1 - Inf -> clipped to 0: the BEST score
    for e in (-38, -20, 0, 20, 38):                               # rows whose largest magnitude stays within [2^-40, 2^40]
        s = np.float32(2.0 ** e)
        scaled = (base * s).astype(np.float32)
        assert orc.cosine(q, scaled, accurate=True) == true       # bit for bit
        assert abs(orc.cosine(q, scaled, accurate=False) - true) < 1e-5
        assert orc.cosine((q * s).astype(np.float32), scaled, accurate=True) == true


def test_a_zero_query_scores_every_point_zero_in_the_store():
    """Store::search_line_embeddings with an all-zero query (an empty query, or unknown tokens only: model2vec pools zeros): qdrant's
    cosine_preprocess leaves it as it is, every dot product is 0 -> distance 1.0 for EVERY point, zero rows included (search_documents'
    simsimd rule says (zero, zero) -> 0 instead).  With a threshold nothing passes unless 0 > 1 - max_distance; without one the first
    top_k rows of the subset come back in storage order.  The library answers such a query with this constant
    (semtools_amd/csrc/search.cpp workspace_zero_query_hits; tests/test_gpu_small_calls.py)."""
    emb = synth.unit_rows(200, seed=3, dup_frac=0, zero_frac=0.05)
    row_path = (np.arange(200) >= 50).astype(np.uint32)
    row_line = np.arange(200, dtype=np.int32)
    q = np.zeros(256, dtype=np.float32)
    assert orc.search_line_embeddings(emb, row_path, row_line, q, [1], 4, 0.5) == []
    res = orc.search_line_embeddings(emb, row_path, row_line, q, [1], 4, None)
    assert [r["row"] for r in res] == [50, 51, 52, 53] and all(r["distance"] == 1.0 for r in res)
    res = orc.search_line_embeddings(emb, row_path, row_line, q, [1], 4, 1.5)
    assert [r["row"] for r in res] == [50, 51, 52, 53]
    assert orc.cosine(q, np.zeros(256, dtype=np.float32), accurate=True) == 0.0       # (search_documents' rule, for contrast)
