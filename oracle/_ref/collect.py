#!/usr/bin/env python3
"""Step 3 of the pinning recipe (oracle/_ref/README.md): pack what the real reference returned (oracle/_ref/out/, written
by `cargo run` of src/main.rs) together with the inputs it was run on into the committed fixtures

    tests/golden/ref_embed.npz     model2vec-rs 0.1.3: lines / queries -> embeddings (2048 / 512 / 4-token caps, encode_single)
    tests/golden/ref_search.npz    simsimd 6.5.1 cosine matrix + semtools v3.0.0 search_documents hits
    tests/golden/ref_store.npz     semtools v3.0.0 Store::search_line_embeddings hits (qdrant-edge)

tests/test_oracle.py compares the C oracle with them when they exist and reports "parity unpinned" while they do not.
Run from the repo root after steps 1 and 2:  python oracle/_ref/collect.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    inp, out = os.path.join(HERE, "inputs"), os.path.join(HERE, "out")
    man = json.load(open(os.path.join(inp, "manifest.json")))
    versions = json.load(open(os.path.join(out, "versions.json")))
    lines = open(os.path.join(inp, "lines.txt")).read().split("\n")[:-1]
    queries = open(os.path.join(inp, "queries.txt")).read().split("\n")[:-1]
    f32 = lambda name, shape: np.fromfile(os.path.join(out, name), dtype="<f4").reshape(shape)  # noqa: E731
    n, nq = man["n_lines"], man["n_queries_text"]
    np.savez_compressed(os.path.join(GOLD, "ref_embed.npz"), V=man["V"], table_seed=man["table_seed"],
                        lines=np.array(lines, dtype=object), queries=np.array(queries, dtype=object),
                        emb_2048=f32("embed_lines_2048.f32", (n, 256)), emb_512=f32("embed_lines_512.f32", (n, 256)),
                        emb_4=f32("embed_lines_4.f32", (n, 256)), emb_single=f32("embed_queries_single.f32", (nq, 256)),
                        versions=json.dumps(versions), allow_pickle=True)
    corpus = np.fromfile(os.path.join(inp, "corpus.f32"), dtype="<f4").reshape(-1, 256)
    qs = np.fromfile(os.path.join(inp, "queries.f32"), dtype="<f4").reshape(-1, 256)
    cos = np.fromfile(os.path.join(out, "simsimd_cosine_f64.bin"), dtype="<f8").reshape(len(qs), len(corpus))
    np.savez_compressed(os.path.join(GOLD, "ref_search.npz"), corpus=corpus, queries=qs, simsimd_cosine=cos,
                        search_documents=open(os.path.join(out, "search_documents.json")).read(), versions=json.dumps(versions))
    store_rows = np.fromfile(os.path.join(inp, "store_rows.f32"), dtype="<f4").reshape(-1, 256)
    np.savez_compressed(os.path.join(GOLD, "ref_store.npz"), store_rows=store_rows, queries=qs,
                        store_search=open(os.path.join(out, "store_search.json")).read(), versions=json.dumps(versions))
    print("wrote", [f for f in sorted(os.listdir(GOLD)) if f.startswith("ref_")])


if __name__ == "__main__":
    sys.exit(main())
