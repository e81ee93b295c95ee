"""Object wrappers over the C ABI handles (context / model / corpus)."""
import ctypes as C
import math

import numpy as np

from . import _lib as L


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Context:
    """One GPU + one HIP stream (smt_ctx).

    stream=None: the library creates a private non-blocking stream.
    stream=<int>: raw hipStream_t to enqueue on (0 = the null stream), e.g.
    torch.cuda.current_stream().cuda_stream."""

    def __init__(self, device=0, stream=None, _borrowed=None):
        self._h = C.c_void_p()
        self._owned = _borrowed is None
        if _borrowed is not None:          # a context owned by a Group (smt_group_ctx)
            self._h = C.c_void_p(_borrowed)
        elif stream is None:
            L.check(L.lib().smt_ctx_create(int(device), C.byref(self._h)))
        else:
            L.check(L.lib().smt_ctx_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(self._h)))
        self.device = device

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if getattr(self, "_owned", True):
                L.lib().smt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        L.check(L.lib().smt_ctx_synchronize(self._h))

    def uncertain_count(self, reset=True):
        """Selects whose exactness certificate failed since the last reset (device entry points only count them)."""
        n = C.c_uint64()
        L.check(L.lib().smt_ctx_uncertain_count(self._h, C.byref(n), int(bool(reset))))
        return int(n.value)

    def deliveries(self):
        """Host-form searches whose answer the select kernel delivered into pinned host memory (smt_debug_deliveries)."""
        n = C.c_uint64(0)
        L.check(L.lib().smt_debug_deliveries(self._h, C.byref(n)))
        return int(n.value)

    def aux_stream(self):
        """Raw hipStream_t of the context's second stream (async selects run there); wrap it with
        torch.cuda.ExternalStream to chain torch / RCCL work behind an async select."""
        st = C.c_void_p()
        L.check(L.lib().smt_ctx_aux_stream(self._h, C.byref(st)))
        return int(st.value or 0)

    def set_tuning(self, key, value):
        L.check(L.lib().smt_set_tuning(self._h, key.encode(), int(value)))

    def prof_enable(self, on=True):
        L.check(L.lib().smt_prof_enable(self._h, int(bool(on))))

    def prof_reset(self):
        L.check(L.lib().smt_prof_reset(self._h))

    def prof_read(self, kernel):
        n, ms = C.c_uint64(), C.c_double()
        L.check(L.lib().smt_prof_read(self._h, kernel.encode(), C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def merge_topk_device(self, rows_ptr, dist_ptr, n_lists, nq, k_in, k_out, out_rows_ptr, out_dist_ptr):
        L.check(L.lib().smt_merge_topk_device(self._h, C.c_void_p(rows_ptr), C.c_void_p(dist_ptr), n_lists, nq,
                                              k_in, k_out, C.c_void_p(out_rows_ptr), C.c_void_p(out_dist_ptr)))


    def merge_topk_packed_device(self, packed_ptr, n_lists, nq, k_in, k_out, out_packed_ptr):
        L.check(L.lib().smt_merge_topk_packed_device(self._h, C.c_void_p(packed_ptr), n_lists, nq, k_in, k_out,
                                                     C.c_void_p(out_packed_ptr)))


class Model:
    """Device-resident model2vec table (smt_model)."""

    def __init__(self, ctx, table=None, normalize=True, device_ptr=None, V=None):
        self.ctx = ctx
        self._h = C.c_void_p()
        if device_ptr is not None:
            L.check(L.lib().smt_model_create_from_device(ctx._h, C.c_void_p(device_ptr), int(V), L.DIM,
                                                         int(normalize), C.byref(self._h)))
            self.V = int(V)
        else:
            table = _f32c(table)
            assert table.ndim == 2
            L.check(L.lib().smt_model_create(ctx._h, L.np_ptr(table), table.shape[0], table.shape[1],
                                             int(normalize), C.byref(self._h)))
            self.V = table.shape[0]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            # a handle that outlived its Context (an error path, garbage collection at interpreter exit) is dropped, not
            # destroyed: smt_model_destroy would read the freed context (the C ABI's rule is children first)
            if getattr(getattr(self, "ctx", None), "_h", None):
                L.lib().smt_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def embed(self, ids, offsets, max_tokens=2048, append_to=None, want_host=True):
        """encode_with_args' pool step for a CSR batch of token ids.  Returns
        ([n_lines x 256] f32 or None, first_row or None)."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = max(offsets.size - 1, 0)
        out = np.empty((n, L.DIM), dtype=np.float32) if want_host else None
        first = C.c_uint64(0)
        L.check(L.lib().smt_embed(self._h, L.np_ptr(ids), L.np_ptr(offsets), n, int(max_tokens),
                                  L.np_ptr(out) if want_host else None,
                                  append_to._h if append_to is not None else None, C.byref(first)))
        return out, (int(first.value) if append_to is not None else None)

    def embed_device(self, ids_ptr, offsets_ptr, n_lines, max_tokens, out_ptr):
        L.check(L.lib().smt_embed_device(self._h, C.c_void_p(ids_ptr), C.c_void_p(offsets_ptr), int(n_lines),
                                         int(max_tokens), C.c_void_p(out_ptr)))


class Corpus:
    """Row-major f32 [rows x 256] matrix resident in HBM (smt_corpus)."""

    def __init__(self, ctx, capacity_rows=0, device_ptr=None, rows=None, _handle=None, _borrowed=False):
        self.ctx = ctx
        self._h = C.c_void_p()
        self._borrowed = _borrowed     # a shard owned by a ShardedCorpus
        if _handle is not None:
            self._h = _handle
        elif device_ptr is not None:
            L.check(L.lib().smt_corpus_from_device(ctx._h, C.c_void_p(device_ptr), int(rows), L.DIM, C.byref(self._h)))
        else:
            L.check(L.lib().smt_corpus_create(ctx._h, L.DIM, int(capacity_rows), C.byref(self._h)))

    @classmethod
    def load(cls, ctx, path):
        h = C.c_void_p()
        L.check(L.lib().smt_corpus_load(ctx._h, str(path).encode(), C.byref(h)))
        return cls(ctx, _handle=h)

    def save(self, path):
        L.check(L.lib().smt_corpus_save(self._h, str(path).encode()))

    def append_to_file(self, path, rows_on_disk):
        L.check(L.lib().smt_corpus_append_to_file(self._h, str(path).encode(), int(rows_on_disk)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if not getattr(self, "_borrowed", False) and getattr(getattr(self, "ctx", None), "_h", None):   # (see Model.close)
                L.lib().smt_corpus_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def rows(self):
        return int(L.lib().smt_corpus_rows(self._h))

    def __len__(self):
        return self.rows

    def append(self, rows):
        rows = _f32c(rows).reshape(-1, L.DIM)
        first = C.c_uint64(0)
        L.check(L.lib().smt_corpus_append_host(self._h, L.np_ptr(rows), rows.shape[0], C.byref(first)))
        return int(first.value)

    def write_rows(self, first_row, rows):
        rows = _f32c(rows).reshape(-1, L.DIM)
        L.check(L.lib().smt_corpus_write_rows(self._h, int(first_row), L.np_ptr(rows), rows.shape[0]))

    def read_rows(self, first_row, n_rows):
        out = np.empty((int(n_rows), L.DIM), dtype=np.float32)
        L.check(L.lib().smt_corpus_read_rows(self._h, int(first_row), int(n_rows), L.np_ptr(out)))
        return out

    def truncate(self, n_rows):
        L.check(L.lib().smt_corpus_truncate(self._h, int(n_rows)))

    def prepack(self, enable=True):
        """smt_corpus_prepack: build (or drop) the fp16 operand image the batched searches read -- half the bytes per row.  For a
        corpus adopted from device memory this is the only way to get one; call it again after changing rows."""
        L.check(L.lib().smt_corpus_prepack(self._h, 1 if enable else 0))

    @property
    def image_bytes(self):
        return int(L.lib().smt_corpus_image_bytes(self._h))

    def debug_batched_scores(self, queries, first_row=0, n_rows=None):
        """Test hook (smt_debug_batched_scores): the f32 distances the batched kernels nominate candidates with,
        float32 [n_rows, nq]."""
        q = _f32c(queries).reshape(-1, L.DIM)
        n_rows = self.rows - first_row if n_rows is None else int(n_rows)
        out = np.empty((n_rows, 32), dtype=np.float32)
        L.check(L.lib().smt_debug_batched_scores(self._h, L.np_ptr(q), q.shape[0], int(first_row), n_rows, L.np_ptr(out)))
        return out[:, :q.shape[0]].copy()

    def search(self, queries, top_k=3, max_distance=None, mode=L.MODE_DOCUMENTS, ranges=None, row_base=0,
               out_cap=None):
        """Returns a list (one per query) of (rows uint64[n], dist float64[n]).

        mode/threshold semantics: see include/semtools_hip.h (smt_search)."""
        q = _f32c(queries).reshape(-1, L.DIM)
        nq = q.shape[0]
        if out_cap is None:
            out_cap = max(int(top_k), 1)
            if max_distance is not None and mode == L.MODE_DOCUMENTS:
                out_cap = max(self.rows, 1)
        rng, n_rng = _ranges_arg(ranges)
        while True:
            out_rows = np.empty((nq, out_cap), dtype=np.uint64)
            out_dist = np.empty((nq, out_cap), dtype=np.float64)
            counts = np.zeros(nq, dtype=np.uint64)
            rc = L.lib().smt_search(self._h, L.np_ptr(q), nq, int(top_k),
                                    float("nan") if max_distance is None else float(max_distance), int(mode),
                                    C.cast(rng, C.c_void_p) if rng is not None else None, n_rng, int(row_base),
                                    L.np_ptr(out_rows), L.np_ptr(out_dist), L.np_ptr(counts), int(out_cap))
            if rc == L.SMT_E_TRUNCATED:
                out_cap = int(counts.max())
                continue
            L.check(rc)
            break
        return [(out_rows[i, :int(counts[i])].copy(), out_dist[i, :int(counts[i])].copy()) for i in range(nq)]

    def range_sets(self):
        """(kept, hits, builds) of the range lists this corpus keeps on the device (smt_debug_range_sets)."""
        v = [C.c_uint64() for _ in range(3)]
        L.check(L.lib().smt_debug_range_sets(self._h, *[C.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    def search_topk_device(self, queries_ptr, nq, top_k, row_base, out_rows_ptr, out_dist_ptr, out_status_ptr=None):
        """out_status_ptr: device-addressable uint32[nq] receiving the per-query verdict (smt_search_topk_device_ex: 0 proved exact,
        1 certificate failed, 2 candidate buffer overflowed), or None."""
        if out_status_ptr:
            L.check(L.lib().smt_search_topk_device_ex(self._h, C.c_void_p(queries_ptr), int(nq), int(top_k), int(row_base),
                                                      C.c_void_p(out_rows_ptr), C.c_void_p(out_dist_ptr), C.c_void_p(int(out_status_ptr))))
            return
        L.check(L.lib().smt_search_topk_device(self._h, C.c_void_p(queries_ptr), int(nq), int(top_k), int(row_base),
                                               C.c_void_p(out_rows_ptr), C.c_void_p(out_dist_ptr)))


class IvfPq:
    """IVF-PQ index over a resident Corpus (smt_ivfpq).  Approximate top-k membership, exact distances."""

    def __init__(self, corpus, nlist=4096, train_iters=10, train_sample=0, local_pca=False, _path=None):
        self.corpus = corpus
        self._h = C.c_void_p()
        if _path is not None:
            L.check(L.lib().smt_ivfpq_load(corpus._h, str(_path).encode(), C.byref(self._h)))
            return
        prm = L.SmtIvfPqParams(int(nlist), 32, 8, int(train_iters), int(train_sample), 0, 1 if local_pca else 0)
        L.check(L.lib().smt_ivfpq_build(corpus._h, C.byref(prm), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            L.lib().smt_ivfpq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def save(self, path):
        L.check(L.lib().smt_ivfpq_save(self._h, str(path).encode()))

    def append(self):
        """Take in the rows appended to the corpus since the build (no retraining).  Returns how many."""
        n = C.c_uint64()
        L.check(L.lib().smt_ivfpq_append(self._h, C.byref(n)))
        return int(n.value)

    @classmethod
    def load(cls, corpus, path):
        """Restore an index saved beside `corpus` (fails unless the row count matches)."""
        return cls(corpus, _path=path)

    def info(self):
        n, nl, nb = C.c_uint64(), C.c_uint32(), C.c_uint64()
        ms = (C.c_double * 4)()
        L.check(L.lib().smt_ivfpq_info(self._h, C.byref(n), C.byref(nl), C.byref(nb), ms))
        return dict(rows=int(n.value), nlist=int(nl.value), index_bytes=int(nb.value),
                    build_ms=dict(coarse_kmeans=ms[0], assign_all=ms[1], pq_train=ms[2], sort_encode=ms[3]))

    def list_sizes(self):
        out = np.empty(self.info()["nlist"], dtype=np.uint64)
        L.check(L.lib().smt_ivfpq_list_sizes(self._h, L.np_ptr(out)))
        return out

    def search(self, queries, top_k=10, nprobe=32, rerank=0, row_base=0):
        q = _f32c(queries).reshape(-1, L.DIM)
        nq = q.shape[0]
        cap = max(int(top_k), 1)
        out_rows = np.empty((nq, cap), dtype=np.uint64)
        out_dist = np.empty((nq, cap), dtype=np.float64)
        counts = np.zeros(nq, dtype=np.uint64)
        L.check(L.lib().smt_ivfpq_search(self._h, L.np_ptr(q), nq, int(top_k), int(nprobe), int(rerank), int(row_base),
                                         L.np_ptr(out_rows), L.np_ptr(out_dist), L.np_ptr(counts), cap))
        return [(out_rows[i, :int(counts[i])].copy(), out_dist[i, :int(counts[i])].copy()) for i in range(nq)]


    def search_device(self, queries_ptr, nq, top_k, nprobe, rerank, row_base, out_rows_ptr, out_dist_ptr):
        """Device-resident form (raw pointers; asynchronous on the context's stream)."""
        L.check(L.lib().smt_ivfpq_search_device(self._h, C.c_void_p(queries_ptr), int(nq), int(top_k), int(nprobe), int(rerank),
                                                int(row_base), C.c_void_p(out_rows_ptr), C.c_void_p(out_dist_ptr)))


class PackedRanges:
    """Row ranges marshalled once (a caller that searches the same document subset again and again: bench.py's workspace leg)."""

    def __init__(self, ranges):
        self.arr, self.n = _ranges_arg(list(ranges))


def _ranges_arg(ranges):
    if ranges is None:
        return None, 0
    if isinstance(ranges, PackedRanges):
        return ranges.arr, ranges.n
    n = len(ranges)
    rng = (L.SmtRange * max(n, 1))()
    for i, (b, e) in enumerate(ranges):
        rng[i].begin, rng[i].end = int(b), int(e)
    return rng, n


class Group:
    """A group of GPUs behind the C ABI (smt_group): one context + stream per device and one RCCL communicator.

    Group(devices=[0, 1, ...])                 single process drives every listed GPU (ncclCommInitAll)
    Group.from_rank(device, rank, n, id)       one rank per process; `id` = Group.unique_id() made on rank 0 and
                                               broadcast by the host (see semtools_amd.dist.group_from_torch)"""

    def __init__(self, devices=None, _handle=None):
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            if devices is None:
                n = L.lib().smt_device_count()
                L.check(min(n, 0))
                devices = range(n)
            devs = list(devices)
            arr = (C.c_int * len(devs))(*devs)
            L.check(L.lib().smt_group_create(arr, len(devs), C.byref(self._h)))
        self._ctx = {}

    @classmethod
    def logical(cls, device, n_shards):
        """n logical ranks on one device (peer reads or device copies instead of RCCL): the sharded path on a 1-GPU box."""
        h = C.c_void_p()
        L.check(L.lib().smt_group_create_logical(int(device), int(n_shards), C.byref(h)))
        return cls(_handle=h)

    @classmethod
    def from_ctx(cls, ctx):
        """A one-rank group around an existing Context: every sharded call forwards to its single-GPU counterpart."""
        h = C.c_void_p()
        L.check(L.lib().smt_group_from_ctx(ctx._h, C.byref(h)))
        g = cls(_handle=h)
        g._keep = ctx   # the context must outlive the group
        return g

    @classmethod
    def from_spec(cls, spec):
        """"0,1,2" / "all" / "<device>:<logical shards>" / "<device>" -- what the CLI reads from $SEMTOOLS_DEVICES."""
        h = C.c_void_p()
        L.check(L.lib().smt_host_group_from_spec(str(spec).encode(), C.byref(h)))
        return cls(_handle=h)

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * L.UNIQUE_ID_BYTES)()
        L.check(L.lib().smt_group_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_rank(cls, device, rank, n_ranks, unique_id):
        assert len(unique_id) == L.UNIQUE_ID_BYTES
        h = C.c_void_p()
        buf = (C.c_ubyte * L.UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        L.check(L.lib().smt_group_create_rank(int(device), int(rank), int(n_ranks), buf, C.byref(h)))
        return cls(_handle=h)

    def info(self):
        v = [C.c_int() for _ in range(5)]
        L.check(L.lib().smt_group_info(self._h, *[C.byref(x) for x in v]))
        return dict(n_ranks=v[0].value, n_local=v[1].value, first_rank=v[2].value, rccl_ranks=v[3].value,
                    rccl_version=v[4].value)

    def ctx(self, local_index=0):
        """The library-owned Context of local device i (tuning keys, profiling, Model(...) on that GPU)."""
        if local_index not in self._ctx:
            h = L.lib().smt_group_ctx(self._h, int(local_index))
            if not h:
                raise IndexError(local_index)
            self._ctx[local_index] = Context(_borrowed=h)
        return self._ctx[local_index]

    def synchronize(self):
        L.check(L.lib().smt_group_synchronize(self._h))

    @property
    def transport(self):
        """"peer" (k-lists read in place by the merging device: the default of one-process groups), "rccl" or "copy"."""
        t = L.lib().smt_group_transport(self._h)
        L.check(min(t, 0))
        return L.TRANSPORT_NAMES[t]

    def set_transport(self, name):
        code = {v: k for k, v in L.TRANSPORT_NAMES.items()}[name]
        L.check(L.lib().smt_group_set_transport(self._h, code))

    def barrier(self):
        L.check(L.lib().smt_group_barrier(self._h))

    def debug_fail_next(self, where, code):
        """Test hook (smt_debug_group_fail_next): the next local step of kind `where` on this process's ranks fails with `code`."""
        L.check(L.lib().smt_debug_group_fail_next(self._h, int(where), int(code)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            L.lib().smt_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedModel:
    """The embedding table replicated on every GPU of a Group; embed() = Model.embed with the lines dealt over the ranks."""

    def __init__(self, group, table, normalize=True):
        self.group = group
        self._h = C.c_void_p()
        table = _f32c(table).reshape(-1, L.DIM)
        self.V = table.shape[0]
        L.check(L.lib().smt_sharded_model_create(group._h, L.np_ptr(table), self.V, L.DIM, int(normalize), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            L.lib().smt_sharded_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def embed(self, ids, offsets, max_tokens=2048, append_to=None, want_host=True):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        out = np.empty((n, L.DIM), dtype=np.float32) if want_host else None
        first = C.c_uint64(0)
        L.check(L.lib().smt_sharded_embed(self._h, L.np_ptr(ids) if len(ids) else None, L.np_ptr(offsets), n, int(max_tokens),
                                          L.np_ptr(out) if out is not None else None,
                                          append_to._h if append_to is not None else None, C.byref(first)))
        return out, int(first.value)


class ShardedCorpus:
    """Corpus row-sharded over a Group (smt_sharded_corpus): cut into ceil(N / n_ranks)-row ranges when made in one go,
    dealt over the ranks append by append when it grows.  Global row == insertion order either way.
    search() has Corpus.search's arguments and returns the same answer as the unsharded matrix would."""

    def __init__(self, group, rows=None, device_ptrs=None, shard_rows=None, path=None, layout=None, empty=False):
        self.group = group
        self._h = C.c_void_p()
        if empty:
            L.check(L.lib().smt_sharded_corpus_create(group._h, L.DIM, C.byref(self._h)))
        elif path is not None and layout is not None:
            pr = np.ascontiguousarray([p[0] for p in layout], dtype=np.uint64)
            rk = np.ascontiguousarray([p[1] for p in layout], dtype=np.uint32)
            L.check(L.lib().smt_sharded_corpus_load_layout(group._h, str(path).encode(), L.np_ptr(pr), L.np_ptr(rk), len(pr),
                                                           C.byref(self._h)))
        elif path is not None:
            L.check(L.lib().smt_sharded_corpus_load(group._h, str(path).encode(), C.byref(self._h)))
        elif device_ptrs is not None:
            n = len(device_ptrs)
            ptrs = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in device_ptrs])
            cnt = (C.c_uint64 * n)(*[int(r) for r in shard_rows])
            L.check(L.lib().smt_sharded_corpus_from_device(group._h, ptrs, cnt, L.DIM, C.byref(self._h)))
        else:
            rows = _f32c(rows).reshape(-1, L.DIM)
            L.check(L.lib().smt_sharded_corpus_from_host(group._h, L.np_ptr(rows), rows.shape[0], L.DIM, C.byref(self._h)))

    @classmethod
    def load(cls, group, path):
        return cls(group, path=path)

    def save(self, path):
        L.check(L.lib().smt_sharded_corpus_save(self._h, str(path).encode()))

    def append_to_file(self, path, rows_on_disk):
        L.check(L.lib().smt_sharded_corpus_append_to_file(self._h, str(path).encode(), int(rows_on_disk)))

    def layout(self):
        """[(rows, rank), ...] in global row order."""
        n = int(L.lib().smt_sharded_corpus_layout(self._h, None, None, 0))
        pr, rk = np.zeros(max(n, 1), dtype=np.uint64), np.zeros(max(n, 1), dtype=np.uint32)
        L.lib().smt_sharded_corpus_layout(self._h, L.np_ptr(pr), L.np_ptr(rk), n)
        return [(int(pr[i]), int(rk[i])) for i in range(n)]

    def read_rows(self, first_row, n_rows):
        out = np.empty((int(n_rows), L.DIM), dtype=np.float32)
        L.check(L.lib().smt_sharded_corpus_read_rows(self._h, int(first_row), int(n_rows), L.np_ptr(out)))
        return out

    def write_rows(self, first_row, rows):
        rows = _f32c(rows).reshape(-1, L.DIM)
        L.check(L.lib().smt_sharded_corpus_write_rows(self._h, int(first_row), L.np_ptr(rows), rows.shape[0]))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            L.lib().smt_sharded_corpus_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def rows(self):
        return int(L.lib().smt_sharded_corpus_rows(self._h))

    def rank_rows(self):
        out = np.zeros(self.group.info()["n_ranks"], dtype=np.uint64)
        L.check(L.lib().smt_sharded_corpus_rank_rows(self._h, L.np_ptr(out)))
        return out

    def shard(self, local_index=0, want_base=True):
        """(Corpus view of local device i's shard, its first global row, its row count).  A corpus that grew by dealt
        appends has no single base per shard: pass want_base=False (the base comes back as None)."""
        h, base, n = C.c_void_p(), C.c_uint64(), C.c_uint64()
        L.check(L.lib().smt_sharded_corpus_shard(self._h, int(local_index), C.byref(h), C.byref(base) if want_base else None,
                                                 C.byref(n)))
        return Corpus(self.group.ctx(local_index), _handle=h, _borrowed=True), int(base.value) if want_base else None, int(n.value)

    def append(self, rows):
        rows = _f32c(rows).reshape(-1, L.DIM)
        first = C.c_uint64(0)
        L.check(L.lib().smt_sharded_corpus_append_host(self._h, L.np_ptr(rows), rows.shape[0], C.byref(first)))
        return int(first.value)

    def search(self, queries, top_k=3, max_distance=None, mode=L.MODE_DOCUMENTS, ranges=None, out_cap=None):
        q = _f32c(queries).reshape(-1, L.DIM)
        nq = q.shape[0]
        if out_cap is None:
            out_cap = max(int(top_k), 1)
            if max_distance is not None and mode == L.MODE_DOCUMENTS:
                # every row under the threshold: a first guess bounded by a TOTAL budget of 64 MB of result buffers (1000 queries x
                # 65 536 slots x 16 B was 1 GB of numpy arrays up front), grown on SMT_E_TRUNCATED to the true counts
                out_cap = max(int(top_k), min(self.rows, 1 << 16, (64 << 20) // (16 * max(nq, 1))), 1)
        rng, n_rng = _ranges_arg(ranges)
        while True:
            out_rows = np.empty((nq, out_cap), dtype=np.uint64)
            out_dist = np.empty((nq, out_cap), dtype=np.float64)
            counts = np.zeros(nq, dtype=np.uint64)
            rc = L.lib().smt_sharded_search(self._h, L.np_ptr(q), nq, int(top_k),
                                            float("nan") if max_distance is None else float(max_distance), int(mode),
                                            C.cast(rng, C.c_void_p) if rng is not None else None, n_rng,
                                            L.np_ptr(out_rows), L.np_ptr(out_dist), L.np_ptr(counts), int(out_cap))
            if rc == L.SMT_E_TRUNCATED:
                out_cap = int(counts.max())
                continue
            L.check(rc)
            break
        return [(out_rows[i, :int(counts[i])].copy(), out_dist[i, :int(counts[i])].copy()) for i in range(nq)]

    def search_topk_device(self, query_ptrs, nq, top_k, out_packed_ptrs, out_status_ptrs=None):
        """Device-resident form: one queries pointer and one output pointer (or 0/None) per LOCAL device.  out_status_ptrs: per local
        device a uint32[nq] buffer (or 0/None) for the per-query verdicts -- the worst status over the shards (_ex entry point)."""
        n = len(query_ptrs)
        qp = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in query_ptrs])
        op = (C.c_void_p * n)(*[C.c_void_p(int(p)) if p else C.c_void_p(None) for p in out_packed_ptrs])
        if out_status_ptrs is not None:
            sp = (C.c_void_p * n)(*[C.c_void_p(int(p)) if p else C.c_void_p(None) for p in out_status_ptrs])
            L.check(L.lib().smt_sharded_search_topk_device_ex(self._h, qp, int(nq), int(top_k), op, sp))
            return
        L.check(L.lib().smt_sharded_search_topk_device(self._h, qp, int(nq), int(top_k), op))


class ShardedIvfPq:
    """IVF index over a ShardedCorpus (smt_sharded_ivfpq): every rank indexes its rows; shared_centroids=True runs
    the coarse k-means data-parallel with an all-reduce of the centroid sums, so all ranks share one set of lists."""

    def __init__(self, sharded_corpus, nlist=4096, train_iters=10, local_pca=True, shared_centroids=True, _path=None):
        self.corpus = sharded_corpus
        self._h = C.c_void_p()
        if _path is not None:
            L.check(L.lib().smt_sharded_ivfpq_load(sharded_corpus._h, str(_path).encode(), C.byref(self._h)))
            return
        prm = L.SmtIvfPqParams(int(nlist), 32, 8, int(train_iters), 0, 0, 1 if local_pca else 0)
        L.check(L.lib().smt_sharded_ivfpq_build(sharded_corpus._h, C.byref(prm), int(bool(shared_centroids)), C.byref(self._h)))

    @classmethod
    def load(cls, sharded_corpus, path):
        return cls(sharded_corpus, _path=path)

    def save(self, path):
        L.check(L.lib().smt_sharded_ivfpq_save(self._h, str(path).encode()))

    def append(self):
        n = C.c_uint64(0)
        L.check(L.lib().smt_sharded_ivfpq_append(self._h, C.byref(n)))
        return int(n.value)

    def info(self):
        rows, nlist, nbytes = C.c_uint64(), C.c_uint32(), C.c_uint64()
        L.check(L.lib().smt_sharded_ivfpq_info(self._h, C.byref(rows), C.byref(nlist), C.byref(nbytes)))
        return dict(rows=int(rows.value), nlist=int(nlist.value), index_bytes=int(nbytes.value))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            L.lib().smt_sharded_ivfpq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shard_list_sizes(self, local_index, nlist):
        h = L.lib().smt_sharded_ivfpq_shard(self._h, int(local_index))
        out = np.empty(int(nlist), dtype=np.uint64)
        L.check(L.lib().smt_ivfpq_list_sizes(C.c_void_p(h), L.np_ptr(out)))
        return out

    def search(self, queries, top_k=10, nprobe=16, rerank=128):
        q = _f32c(queries).reshape(-1, L.DIM)
        nq = q.shape[0]
        cap = max(int(top_k), 1)
        out_rows = np.empty((nq, cap), dtype=np.uint64)
        out_dist = np.empty((nq, cap), dtype=np.float64)
        counts = np.zeros(nq, dtype=np.uint64)
        L.check(L.lib().smt_sharded_ivfpq_search(self._h, L.np_ptr(q), nq, int(top_k), int(nprobe), int(rerank),
                                                 L.np_ptr(out_rows), L.np_ptr(out_dist), L.np_ptr(counts), cap))
        return [(out_rows[i, :int(counts[i])].copy(), out_dist[i, :int(counts[i])].copy()) for i in range(nq)]


def merge_topk(rows, dist, k_out):
    """Host merge of per-shard sorted top-k lists laid out [n_lists][nq][k_in]."""
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    dist = np.ascontiguousarray(dist, dtype=np.float64)
    n_lists, nq, k_in = rows.shape
    out_rows = np.empty((nq, k_out), dtype=np.uint64)
    out_dist = np.empty((nq, k_out), dtype=np.float64)
    counts = np.zeros(nq, dtype=np.uint64)
    L.check(L.lib().smt_merge_topk(L.np_ptr(rows), L.np_ptr(dist), n_lists, nq, k_in, int(k_out),
                                   L.np_ptr(out_rows), L.np_ptr(out_dist), L.np_ptr(counts)))
    return out_rows, out_dist, counts
