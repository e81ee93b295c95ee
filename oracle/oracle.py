"""ctypes binding of oracle/libsemtools_oracle.so (C restatement).

TEST INFRASTRUCTURE ONLY -- see oracle/semtools_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsemtools_oracle.so")


def build(force=False):
    """Compile the oracle with oracle/Makefile (gcc only)."""
    srcs = [os.path.join(_HERE, f) for f in ("semtools_oracle.c", "cpu_fast.c", "semtools_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean"])
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class OrcResult(C.Structure):
    _fields_ = [("doc", C.c_uint64), ("match_line", C.c_uint64), ("start", C.c_uint64),
                ("end", C.c_uint64), ("distance", C.c_double)]


class OrcRankedLine(C.Structure):
    _fields_ = [("path_id", C.c_uint32), ("line_number", C.c_int32), ("distance", C.c_float),
                ("row", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        f32p, u32p, u64p, i32p, f64p = (C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_double))
        L.orc_pool_ids.argtypes = [f32p, C.c_uint64, C.c_uint32, C.c_int, u32p, C.c_uint64, C.c_uint32, f32p]
        L.orc_pool_ids.restype = None
        L.orc_embed_lines.argtypes = [f32p, C.c_uint64, C.c_uint32, C.c_int, u32p, u64p, C.c_uint64, C.c_uint32, f32p]
        L.orc_embed_lines.restype = None
        for fn in (L.orc_cosine_f32_serial, L.orc_cosine_f32_accurate):
            fn.argtypes = [f32p, f32p, C.c_uint32]
            fn.restype = C.c_double
        L.orc_search_documents.argtypes = [f32p, u64p, C.c_uint64, C.c_uint32, f32p, C.c_uint64, C.c_uint64,
                                           C.c_int, C.c_double, C.c_int, C.POINTER(OrcResult), C.c_uint64]
        L.orc_search_documents.restype = C.c_uint64
        L.orc_search_line_embeddings.argtypes = [f32p, u32p, i32p, C.c_uint64, C.c_uint32, f32p, u32p, C.c_uint64,
                                                 C.c_uint64, C.c_int, C.c_float, C.POINTER(OrcRankedLine), C.c_uint64]
        L.orc_search_line_embeddings.restype = C.c_uint64
        L.orc_fnv1a_hash.argtypes = [C.c_char_p, C.c_uint64]
        L.orc_fnv1a_hash.restype = C.c_uint64
        L.orc_line_embedding_id.argtypes = [C.c_char_p, C.c_int32]
        L.orc_line_embedding_id.restype = C.c_uint64
        L.orc_doc_meta_id.argtypes = [C.c_char_p]
        L.orc_doc_meta_id.restype = C.c_uint64
        L.orc_scan_topk_threads.argtypes = [f32p, C.c_uint64, C.c_uint32, f32p, C.c_uint64, C.c_int, u64p, f64p]
        L.orc_scan_topk_threads.restype = C.c_uint64
        L.orc_search_documents_simd.argtypes = [f32p, C.c_uint64, C.c_uint32, f32p, C.c_uint64, C.c_uint64, C.c_int,
                                                C.c_double, C.POINTER(OrcResult), C.c_uint64]
        L.orc_search_documents_simd.restype = C.c_uint64
        L.orc_simd_backend.restype = C.c_char_p
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def pool_ids(table, ids, normalize=True, max_tokens=0):
    table = _f32(table)
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    out = np.empty(table.shape[1], dtype=np.float32)
    lib().orc_pool_ids(_p(table, C.c_float), table.shape[0], table.shape[1], int(normalize),
                       _p(ids, C.c_uint32), ids.size, max_tokens, _p(out, C.c_float))
    return out


def embed_lines(table, ids, offsets, normalize=True, max_tokens=0):
    table = _f32(table)
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    out = np.empty((n, table.shape[1]), dtype=np.float32)
    lib().orc_embed_lines(_p(table, C.c_float), table.shape[0], table.shape[1], int(normalize),
                          _p(ids, C.c_uint32), _p(offsets, C.c_uint64), n, max_tokens, _p(out, C.c_float))
    return out


def cosine(a, b, accurate=False):
    a, b = _f32(a), _f32(b)
    assert a.size == b.size  # simsimd returns None on mismatch (src/search/mod.rs:87)
    fn = lib().orc_cosine_f32_accurate if accurate else lib().orc_cosine_f32_serial
    return fn(_p(a, C.c_float), _p(b, C.c_float), a.size)


def search_documents(emb, doc_line_counts, query, n_lines=3, top_k=3, max_distance=None, accurate=False):
    """A6.  Returns list of dicts(doc, match_line, start, end, distance)."""
    emb = _f32(emb).reshape(-1, len(query)) if np.size(emb) else np.zeros((0, len(query)), np.float32)
    counts = np.ascontiguousarray(doc_line_counts, dtype=np.uint64)
    q = _f32(query)
    total = int(counts.sum()) if counts.size else 0
    assert total == emb.shape[0]
    cap = max(total, 1)
    out = (OrcResult * cap)()
    n = lib().orc_search_documents(_p(emb, C.c_float), _p(counts, C.c_uint64), counts.size, q.size,
                                   _p(q, C.c_float), n_lines, top_k, int(max_distance is not None),
                                   float(max_distance if max_distance is not None else 0.0), int(accurate), out, cap)
    return [dict(doc=int(r.doc), match_line=int(r.match_line), start=int(r.start), end=int(r.end),
                 distance=float(r.distance)) for r in out[:n]]


def search_line_embeddings(emb, row_path, row_line, query, subset, top_k, max_distance=None):
    """A10.  Returns list of dicts(path_id, line_number, distance(f32), row)."""
    q = _f32(query)
    emb = _f32(emb).reshape(-1, q.size)
    row_path = np.ascontiguousarray(row_path, dtype=np.uint32)
    row_line = np.ascontiguousarray(row_line, dtype=np.int32)
    subset = np.ascontiguousarray(subset, dtype=np.uint32)
    cap = max(int(top_k), 1)
    out = (OrcRankedLine * cap)()
    n = lib().orc_search_line_embeddings(_p(emb, C.c_float), _p(row_path, C.c_uint32), _p(row_line, C.c_int32),
                                         emb.shape[0], q.size, _p(q, C.c_float), _p(subset, C.c_uint32),
                                         subset.size, top_k, int(max_distance is not None),
                                         float(max_distance if max_distance is not None else 0.0), out, cap)
    return [dict(path_id=int(r.path_id), line_number=int(r.line_number), distance=float(np.float32(r.distance)),
                 row=int(r.row)) for r in out[:n]]


def fnv1a_hash(b: bytes):
    return lib().orc_fnv1a_hash(b, len(b))


def line_embedding_id(path: str, line_number: int):
    return lib().orc_line_embedding_id(path.encode(), line_number)


def doc_meta_id(path: str):
    return lib().orc_doc_meta_id(path.encode())


def scan_topk_threads(emb, query, top_k, n_threads):
    q = _f32(query)
    emb = _f32(emb).reshape(-1, q.size)
    rows = np.empty(top_k, dtype=np.uint64)
    dist = np.empty(top_k, dtype=np.float64)
    n = lib().orc_scan_topk_threads(_p(emb, C.c_float), emb.shape[0], q.size, _p(q, C.c_float), top_k,
                                    n_threads, _p(rows, C.c_uint64), _p(dist, C.c_double))
    return rows[:n], dist[:n]


def search_documents_simd(emb, query, n_lines=3, top_k=3, max_distance=None):
    """The reference's control flow with a SIMD cosine, single thread (bench.py's "port-simd" baseline)."""
    q = _f32(query)
    emb = _f32(emb).reshape(-1, q.size)
    cap = max(int(top_k) if max_distance is None else emb.shape[0], 1)
    out = (OrcResult * cap)()
    n = lib().orc_search_documents_simd(_p(emb, C.c_float), emb.shape[0], q.size, _p(q, C.c_float), n_lines, top_k,
                                        int(max_distance is not None),
                                        float(max_distance if max_distance is not None else 0.0), out, cap)
    return [dict(doc=0, match_line=int(r.match_line), start=int(r.start), end=int(r.end), distance=float(r.distance))
            for r in out[:min(n, cap)]]


def simd_backend():
    return lib().orc_simd_backend().decode()
