"""Independent numpy twin of the C oracle -- TEST INFRASTRUCTURE ONLY.

Written separately from semtools_oracle.c so that the two restatements can
cross-check each other: pooling must agree bit-for-bit,
distances to <= 1e-7.  Same citations as the C file.
"""
import numpy as np

F = np.float32


def pool_ids(table, ids, normalize=True, max_tokens=0):
    """model2vec-rs pool_ids [UPSTREAM-RECALL]; call site src/search/mod.rs:69."""
    ids = np.asarray(ids, dtype=np.int64)
    if max_tokens:
        ids = ids[:max_tokens]
    D = table.shape[1]
    acc = np.zeros(D, dtype=F)
    for i in ids:                       # token order, f32 adds, per-dimension chains
        if i < table.shape[0]:
            acc = (acc + table[i].astype(F)).astype(F)
    cnt = F(max(len(ids), 1))
    acc = (acc / cnt).astype(F)
    if normalize:
        sq = (acc * acc).astype(F)
        ss = np.add.accumulate(sq, dtype=F)[-1] if D else F(0)   # sequential f32 sum
        norm = max(F(np.sqrt(ss, dtype=F)), F(1e-12))
        acc = (acc / norm).astype(F)
    return acc


def _seq_sum(x):
    return np.add.accumulate(x.astype(F), dtype=F)[-1] if x.size else F(0)


def cosine_serial(a, b):
    """simsimd serial f32 cosine [UPSTREAM-RECALL]; call site src/search/mod.rs:86."""
    a = np.asarray(a, F)
    b = np.asarray(b, F)
    ab, a2, b2 = _seq_sum(a * b), _seq_sum(a * a), _seq_sum(b * b)
    return _finish(float(ab), float(a2), float(b2))


def cosine_accurate(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    ab = np.add.accumulate(a * b)[-1]
    a2 = np.add.accumulate(a * a)[-1]
    b2 = np.add.accumulate(b * b)[-1]
    return _finish(float(ab), float(a2), float(b2))


def _finish(ab, a2, b2):
    if a2 == 0 and b2 == 0:
        return 0.0
    if ab == 0:
        return 1.0
    d = 1.0 - ab * (1.0 / np.sqrt(a2)) * (1.0 / np.sqrt(b2))
    return d if d > 0 else 0.0


def search_documents(docs_emb, query, n_lines=3, top_k=3, max_distance=None, accurate=False):
    """src/search/mod.rs:77-120.  docs_emb: list of [len_i x D] arrays."""
    cos = cosine_accurate if accurate else cosine_serial
    res = []
    thr = 100.0 if max_distance is None else max_distance
    for d, emb in enumerate(docs_emb):
        n = len(emb)
        for idx in range(n):
            dist = cos(query, emb[idx])
            if dist < thr:
                res.append(dict(doc=d, match_line=idx, start=max(0, idx - n_lines),
                                end=min(n, idx + n_lines + 1), distance=dist))
    res.sort(key=lambda r: r["distance"])  # list.sort is stable
    return res if max_distance is not None else res[:top_k]


def fnv1a_hash(b: bytes):
    """src/workspace/store.rs:651-661."""
    h = 0xCBF29CE484222325
    for x in b:
        h ^= x
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def line_embedding_id(path: str, line_number: int):
    """src/workspace/store.rs:82-89."""
    return fnv1a_hash(path.encode() + int(line_number).to_bytes(4, "little", signed=True))
