"""Loading a real model2vec model directory (tokenizer.json + model.safetensors + config.json) for the host layer.

Tokenisation uses the Hugging Face `tokenizers` package -- the same Rust library the reference reaches through
model2vec-rs -- plugged into the C++ host layer as a callback tokenizer; the embedding table is uploaded once.
(The hub download of StaticModel::from_pretrained is out of scope: no network here.  Point this at a local copy
of minishlab/potion-multilingual-128M and the pipeline is the reference's: encode_batch(add_special_tokens=False)
-> drop unk -> truncate -> pool on the GPU.)"""
import json
import os

import numpy as np

from . import host


def load_static_model(ctx, model_dir):
    from safetensors.numpy import load_file
    from tokenizers import Tokenizer

    tok = Tokenizer.from_file(os.path.join(model_dir, "tokenizer.json"))
    tensors = load_file(os.path.join(model_dir, "model.safetensors"))
    emb = np.ascontiguousarray(tensors["embeddings"].astype(np.float32))
    normalize = True
    cfg_path = os.path.join(model_dir, "config.json")
    if os.path.exists(cfg_path):
        normalize = bool(json.load(open(cfg_path)).get("normalize", True))
    spec = json.loads(tok.to_str())
    unk_token = (spec.get("model") or {}).get("unk_token")
    if unk_token is None and (spec.get("model") or {}).get("unk_id") is not None:      # Unigram stores an index
        unk_id = int(spec["model"]["unk_id"])
    else:
        unk_id = tok.token_to_id(unk_token) if unk_token else None
    vocab = tok.get_vocab()
    lens = sorted(len(t.encode("utf-8")) for t in vocab)                                   # model2vec: median of tk.len() -- BYTES
    median_len = max(1, lens[len(lens) // 2]) if lens else 5

    def encode(text):
        return tok.encode(text, add_special_tokens=False).ids

    return host.StaticModel(ctx, table=emb, tokenizer=encode, normalize=normalize, unk_id=unk_id, median_len=median_len)
