#!/usr/bin/env python3
"""Step 1 of the pinning recipe (oracle/_ref/README.md): write the synthetic inputs the REAL reference is run on.

Everything is generated from the seeded generators in tests/synth.py -- the same inputs the C oracle and the GPU tests see --
into oracle/_ref/inputs/ (not tracked: regenerate it, the bytes are deterministic):

  model/model.safetensors     "embeddings" [V x 256] f32   = synth.table(V, seed=2)
  model/tokenizer.json        WordLevel over w0 .. w{V-2} + [UNK], Whitespace pre-tokenizer (built with the `tokenizers` wheel,
                              i.e. by the same crate model2vec-rs loads it with)
  model/config.json           {"normalize": true}
  lines.txt                   400 lines of pseudo prose incl. an empty line, an all-unknown line and a 3000-token line
  queries.txt                 4 query strings
  corpus.f32 / queries.f32    600 x 256 unit rows (5 % exact duplicates, 1 % zero rows) and 3 unit queries (search_small.npz's)
  store_rows.f32              the three vectors of the reference's own known-answer test (src/workspace/store.rs:814-850) + 60 more
  manifest.json               shapes and seeds

Needs numpy, safetensors, tokenizers (all in the round's image).  Run from the repo root:  python oracle/_ref/make_inputs.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import synth  # noqa: E402

V = 2048


def main():
    from safetensors.numpy import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers

    out = os.path.join(HERE, "inputs")
    os.makedirs(os.path.join(out, "model"), exist_ok=True)
    table = synth.table(V, seed=2)
    save_file({"embeddings": table}, os.path.join(out, "model", "model.safetensors"))
    vocab = {f"w{i}": i for i in range(V - 1)}
    vocab["[UNK]"] = V - 1
    tok = Tokenizer(models.WordLevel(vocab=vocab, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.save(os.path.join(out, "model", "tokenizer.json"))
    with open(os.path.join(out, "model", "config.json"), "w") as f:
        json.dump({"normalize": True}, f)

    lines = synth.pseudo_prose(400, vocab_size=V - 1, seed=1)
    lines[50] = ""                                             # empty line -> zero vector
    lines[60] = "unknownword anotherunknown"                   # only unk ids -> dropped -> zero vector
    lines[70] = " ".join(f"w{(7 * i) % (V - 1)}" for i in range(3000))   # longer than the 2048-token cap
    with open(os.path.join(out, "lines.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    queries = [lines[17], lines[70], "w5 w6 w7", "unknownword"]
    with open(os.path.join(out, "queries.txt"), "w") as f:
        f.write("\n".join(queries) + "\n")

    corpus = synth.unit_rows(600, seed=3, dup_frac=0.05, zero_frac=0.01)
    qs = synth.unit_query(4, nq=3)
    corpus.astype("<f4").tofile(os.path.join(out, "corpus.f32"))
    qs.astype("<f4").tofile(os.path.join(out, "queries.f32"))

    # the store's known-answer vectors: [0.1; 256], [0.5; 256], [0.75; 256] (store.rs:814-850), then 60 generic rows
    store_rows = np.concatenate([np.full((1, 256), 0.1, np.float32), np.full((1, 256), 0.5, np.float32),
                                 np.full((1, 256), 0.75, np.float32), synth.unit_rows(60, seed=9, dup_frac=0.0, zero_frac=0.0) * 3.0])
    store_rows.astype("<f4").tofile(os.path.join(out, "store_rows.f32"))
    with open(os.path.join(out, "manifest.json"), "w") as f:
        json.dump({"V": V, "dim": 256, "n_lines": len(lines), "n_queries_text": len(queries), "corpus_rows": 600, "n_queries_vec": 3,
                   "store_rows": int(store_rows.shape[0]), "table_seed": 2, "corpus_seed": 3, "query_seed": 4}, f, indent=1)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
