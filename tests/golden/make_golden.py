"""Generates tests/golden/*.npz -- small pinned input/output vectors for the hot path.

The reference itself (Rust + un-vendored crates, no cargo here) cannot be run, and its own
tests hold NO numeric golden vectors for this path.  These fixtures are
therefore produced by the C oracle (oracle/semtools_oracle.c) and accepted only if the
independent numpy twin (oracle/oracle_np.py) reproduces them: embeddings bit-for-bit,
distances to <= 1e-7, indices exactly.  Run from the repo root:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc, oracle_np as onp  # noqa: E402
from tests import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    # ---- embed: 64 ragged lines over a 512-row table
    table = synth.table(512, seed=2)
    ids, offsets = synth.token_lines(64, V=512, seed=1, min_tok=0, max_tok=12)
    emb = orc.embed_lines(table, ids, offsets, True, 2048)
    for i in range(64):
        twin = onp.pool_ids(table, ids[int(offsets[i]): int(offsets[i + 1])], True, 2048)
        assert np.array_equal(emb[i], twin), i
    emb_cap = orc.embed_lines(table, ids, offsets, True, 4)
    np.savez_compressed(os.path.join(OUT, "embed_small.npz"), table_seed=2, V=512, ids=ids, offsets=offsets,
                        emb=emb, emb_cap4=emb_cap)

    # ---- search: 600 rows (dups + zero rows), 3 queries, top-k and threshold
    corpus = synth.unit_rows(600, seed=3, dup_frac=0.05, zero_frac=0.01)
    qs = synth.unit_query(4, nq=3)
    cases = {}
    for qi in range(3):
        for k in (1, 3, 10):
            for acc in (False, True):
                res = orc.search_documents(corpus, [600], qs[qi], n_lines=3, top_k=k, accurate=acc)
                twin = onp.search_documents([corpus], qs[qi], 3, k, None, acc)
                assert [r["match_line"] for r in res] == [r["match_line"] for r in twin]
                assert np.allclose([r["distance"] for r in res], [r["distance"] for r in twin], rtol=0, atol=1e-7)
                cases[f"q{qi}_k{k}_{'acc' if acc else 'ser'}_rows"] = np.array([r["match_line"] for r in res], np.int64)
                cases[f"q{qi}_k{k}_{'acc' if acc else 'ser'}_dist"] = np.array([r["distance"] for r in res])
        res = orc.search_documents(corpus, [600], qs[qi], n_lines=3, top_k=3, max_distance=0.9, accurate=True)
        twin = onp.search_documents([corpus], qs[qi], 3, 3, 0.9, True)
        assert [r["match_line"] for r in res] == [r["match_line"] for r in twin]
        cases[f"q{qi}_thr0.9_rows"] = np.array([r["match_line"] for r in res], np.int64)
        cases[f"q{qi}_thr0.9_dist"] = np.array([r["distance"] for r in res])
    np.savez_compressed(os.path.join(OUT, "search_small.npz"), corpus=corpus, queries=qs, **cases)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
