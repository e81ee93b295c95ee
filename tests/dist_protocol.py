"""The exchange step of the N > 1 path restated over torch.distributed CPU tensors ("gloo"): TEST INFRASTRUCTURE.

On GPUs the exchange lives inside libsemtools_hip.so (csrc/group.cpp: per-shard scan -> ncclAllGather of the packed k-lists ->
merge_topk_kernel; threshold mode: counts, then one padded gather).  A box without GPUs cannot run that; what it CAN run is the
protocol -- padding, packed [nq][2][k] layout, threshold mode's count-then-padded-gather, the (distance, global row) merge -- over
gloo with world_size 2, with the CPU oracle standing in for the per-shard scan (tests/test_dist_cpu.py).  Until round 4 this lived
in semtools_amd/dist.py beside the library-side group that replaced it on the product path; nothing in the product imports it.
"""
import numpy as np
import torch
import torch.distributed as dist

from semtools_amd import core
from semtools_amd._lib import MODE_DOCUMENTS
from semtools_amd.dist import shard_bounds  # noqa: F401  (the tests cut shards the way the library does)

PAD_ROW = -1  # UINT64_MAX viewed as int64

def allgather_merge_packed(local_packed, k_out, ctx=None, group=None, gathered=None, out=None):
    """The one-collective form: local_packed int64 [nq, 2, k] = (row bit patterns, float64 distance bit
    patterns) exactly as smt_search_topk_device wrote them into ONE buffer.  A single all-gather moves
    both; the merge kernel (or the host merge on CPU tensors) reads the packed layout directly.
    Returns int64 [nq, 2, k_out] (rows in [:, 0], distance bits in [:, 1])."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    nq, two, k_in = local_packed.shape
    assert two == 2 and local_packed.dtype == torch.int64 and local_packed.is_contiguous()
    g = gathered if gathered is not None else torch.empty((world, nq, 2, k_in), dtype=torch.int64,
                                                          device=local_packed.device)
    if world > 1:
        dist.all_gather_into_tensor(g.view(world * nq * 2, k_in), local_packed.view(nq * 2, k_in), group=group)
    else:
        g[0].copy_(local_packed)
    o = out if out is not None else torch.empty((nq, 2, k_out), dtype=torch.int64, device=local_packed.device)
    assert not local_packed.is_cuda, "CPU protocol model: the device exchange is the library's (csrc/group.cpp)"
    rows_u = np.ascontiguousarray(g[:, :, 0, :].numpy()).view(np.uint64)
    dd = np.ascontiguousarray(g[:, :, 1, :].numpy()).view(np.float64)
    mr, md, _ = core.merge_topk(rows_u, dd, k_out)
    o[:, 0, :] = torch.from_numpy(mr.view(np.int64))
    o[:, 1, :] = torch.from_numpy(md.view(np.int64))
    return o


def allgather_merge_topk(local_rows, local_dist, k_out, ctx=None, group=None, gathered=None, out=None):
    """local_rows int64 [nq,k] (uint64 bit pattern, padding = -1), local_dist float64 [nq,k].

    Returns (rows int64 [nq,k_out], dist float64 [nq,k_out]) identical on every rank.
    CUDA tensors: RCCL all-gather + device merge kernel on `ctx`'s stream (pass the
    smt Context created on torch's current stream).  CPU tensors: gloo + host merge."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    nq, k_in = local_rows.shape
    if gathered is None:
        g_rows = torch.empty((world, nq, k_in), dtype=torch.int64, device=local_rows.device)
        g_dist = torch.empty((world, nq, k_in), dtype=torch.float64, device=local_rows.device)
    else:
        g_rows, g_dist = gathered
    if world > 1:
        # flat [world*nq, k] views: the layout is [rank][query][k] either way
        dist.all_gather_into_tensor(g_rows.view(world * nq, k_in), local_rows.contiguous(), group=group)
        dist.all_gather_into_tensor(g_dist.view(world * nq, k_in), local_dist.contiguous(), group=group)
    else:
        g_rows[0].copy_(local_rows)
        g_dist[0].copy_(local_dist)
    assert not local_rows.is_cuda, "CPU protocol model: the device exchange is the library's (csrc/group.cpp)"
    rows_u = g_rows.numpy().view(np.uint64)
    mr, md, _ = core.merge_topk(rows_u, g_dist.numpy(), k_out)
    return torch.from_numpy(mr.view(np.int64)), torch.from_numpy(md)


def _collective_device(group=None):
    return torch.device("cpu")


def allgather_threshold_hits(local_rows, local_dist, top_k=None, group=None):
    """Threshold mode: every rank holds a VARIABLE number of hits (global rows, float64
    distances, sorted distance asc / row asc -- what smt_search returns with max_distance set and
    row_base = the shard's first row).  Exchange = all-gather of the counts, then ONE all-gather of a
    max-count-padded [2, max] int64 buffer (rows, distance bits); every rank then merges redundantly.
    Shards are contiguous ascending row ranges, so the (distance, row) order of the union is the
    single-shard order (src/search/mod.rs:107-111 stable sort == row asc on ties).
    top_k: None = return all hits (search_documents with a threshold, :115-116);
           an int = truncate after the merge (Store::search_line_embeddings, store.rs:543)."""
    rows = np.ascontiguousarray(np.asarray(local_rows, dtype=np.uint64))
    dd = np.ascontiguousarray(np.asarray(local_dist, dtype=np.float64))
    assert rows.shape == dd.shape and rows.ndim == 1
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        n = len(rows) if top_k is None else min(len(rows), top_k)
        return rows[:n].copy(), dd[:n].copy()
    dev = _collective_device(group)
    cnt = torch.tensor([len(rows)], dtype=torch.int64, device=dev)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, cnt, group=group)
    counts = counts.cpu().numpy()
    width = int(counts.max())
    if width == 0:
        return np.empty(0, np.uint64), np.empty(0, np.float64)
    buf = np.zeros((2, width), np.int64)
    buf[0, : len(rows)] = rows.view(np.int64)
    buf[1, : len(rows)] = dd.view(np.int64)
    mine = torch.from_numpy(buf).to(dev)
    g = torch.empty((world * 2, width), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(g, mine, group=group)
    g = g.cpu().numpy().reshape(world, 2, width)
    all_rows = np.concatenate([g[r, 0, : counts[r]] for r in range(world)]).view(np.uint64)
    all_dd = np.concatenate([g[r, 1, : counts[r]] for r in range(world)]).view(np.float64)
    order = np.lexsort((all_rows, all_dd))       # distance asc, then global row asc
    if top_k is not None:
        order = order[:top_k]
    return all_rows[order], all_dd[order]


def exchange_topk(local, top_k, ctx=None, group=None):
    """local: per query (global rows uint64[<=k], float64 distances[<=k]) sorted (distance, row) asc.
    Pads to k, all-gathers (RCCL on device buffers / gloo on host buffers), merges; every rank gets the
    global top-k.  ctx: the smt Context whose device merge kernel is used on the RCCL path."""
    out = []
    nq = len(local)
    rows = np.full((nq, top_k), -1, np.int64)
    dd = np.full((nq, top_k), np.inf)
    for i, (r, d) in enumerate(local):
        rows[i, : len(r)] = np.asarray(r, np.uint64).view(np.int64)
        dd[i, : len(d)] = d
    mr, md = allgather_merge_topk(torch.from_numpy(rows), torch.from_numpy(dd), top_k, group=group)
    mr, md = mr.numpy(), md.numpy()
    for i in range(nq):
        n = int((mr[i] != PAD_ROW).sum())
        out.append((mr[i, :n].view(np.uint64).copy(), md[i, :n].copy()))
    return out


class ShardedCorpus:
    """One rank's view of a row-sharded corpus: the local smt Corpus + where its rows sit globally.

    search() = local smt_search with row_base, then the exchange step above.  Every rank returns the
    same global answer (rows are GLOBAL indices).  With one rank / no process group it degenerates to
    Corpus.search.  The bench drives the device-resident form of the top-k path directly
    (search_topk_device + allgather_merge_packed) to keep the host out of the timed loop; this class
    is the convenience surface with the reference's semantics."""

    def __init__(self, corpus, row_base, group=None):
        self.corpus = corpus
        self.row_base = int(row_base)
        self.group = group

    def search(self, queries, top_k, max_distance=None, mode=MODE_DOCUMENTS):
        q = np.ascontiguousarray(np.asarray(queries, np.float32).reshape(-1, 256))
        local = self.corpus.search(q, top_k=top_k, max_distance=max_distance, mode=mode, row_base=self.row_base)
        threshold_all = max_distance is not None and mode == MODE_DOCUMENTS
        out = []
        if threshold_all or max_distance is not None:
            for rows, dd in local:
                out.append(allgather_threshold_hits(rows, dd, None if threshold_all else top_k, self.group))
            return out
        return exchange_topk(local, top_k, ctx=self.corpus.ctx, group=self.group)
