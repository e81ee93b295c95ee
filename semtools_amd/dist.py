"""Multi-GPU glue for Python hosts: one process per GPU (torchrun), corpus row-sharded, ONE exchange step.

On GPUs the exchange lives INSIDE libsemtools_hip.so (csrc/group.cpp: smt_group_* / smt_sharded_*: per-shard scan ->
ncclAllGather of the packed k-lists -> merge_topk_kernel); this module is a thin caller: `group_from_torch` joins the
ranks of a torch.distributed job into one library group (rank 0's ncclUniqueId travels through torch's store), and
`ShardedCorpus` / `ShardedIvfPq` hand the work to it.  What remains here in Python is the same exchange written over
torch.distributed tensors ("gloo" on CPU tensors), which is how the protocol -- padding, packed layout, threshold
mode's count-then-padded-gather, (distance, row) merge -- is tested on a box without GPUs (tests/test_dist_cpu.py).
There is no reference counterpart (the reference is single-process CPU code); the contract is: sharded result ==
single-shard result (tests/).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import core
from ._lib import MODE_DOCUMENTS

PAD_ROW = -1  # UINT64_MAX viewed as int64


def group_from_torch(device, group=None):
    """One library group (smt_group) spanning the ranks of the current torch.distributed job: rank 0 creates the
    ncclUniqueId, every rank joins with its GPU (ncclCommInitRank inside the library).  Works with any torch
    backend -- the 128 bytes go through broadcast_object_list."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    box = [core.Group.unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    return core.Group.from_rank(int(device), rank, world, box[0])


def shard_bounds(n_rows, world_size):
    """Contiguous row ranges, rows_per_rank = ceil(n / world) (keeps a document's lines together)."""
    per = -(-n_rows // world_size) if world_size > 0 else 0
    return [(min(r * per, n_rows), min((r + 1) * per, n_rows)) for r in range(world_size)]


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
    if local_packed.is_cuda:
        assert ctx is not None, "device merge needs the smt Context bound to torch's current stream"
        ctx.merge_topk_packed_device(g.data_ptr(), world, nq, k_in, k_out, o.data_ptr())
        return o
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
    if local_rows.is_cuda:
        assert ctx is not None, "device merge needs the smt Context bound to torch's current stream"
        if out is None:
            o_rows = torch.empty((nq, k_out), dtype=torch.int64, device=local_rows.device)
            o_dist = torch.empty((nq, k_out), dtype=torch.float64, device=local_rows.device)
        else:
            o_rows, o_dist = out
        ctx.merge_topk_device(g_rows.data_ptr(), g_dist.data_ptr(), world, nq, k_in, k_out,
                              o_rows.data_ptr(), o_dist.data_ptr())
        return o_rows, o_dist
    rows_u = g_rows.numpy().view(np.uint64)
    mr, md, _ = core.merge_topk(rows_u, g_dist.numpy(), k_out)
    return torch.from_numpy(mr.view(np.int64)), torch.from_numpy(md)


def _collective_device(group=None):
    """RCCL moves device buffers, gloo host buffers."""
    if dist.is_initialized() and dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
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
    dev = _collective_device(group)
    if dev.type == "cuda":
        # convenience path: the smt context may own a different stream than torch's current one,
        # so fence both sides (the bench binds the context to torch's stream and needs no fences)
        packed = torch.from_numpy(np.stack([rows, dd.view(np.int64)], axis=1).copy()).to(dev)
        world = dist.get_world_size(group)
        g = torch.empty((world, nq, 2, top_k), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(g.view(world * nq * 2, top_k), packed.view(nq * 2, top_k), group=group)
        torch.cuda.current_stream().synchronize()
        o = torch.empty((nq, 2, top_k), dtype=torch.int64, device=dev)
        ctx.merge_topk_packed_device(g.data_ptr(), world, nq, top_k, top_k, o.data_ptr())
        ctx.synchronize()
        m = o.cpu().numpy()
        mr, md = m[:, 0], np.ascontiguousarray(m[:, 1]).view(np.float64)
    else:
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
        if isinstance(self.corpus, core.ShardedCorpus):     # the library does scan + all-gather + merge itself
            return self.corpus.search(q, top_k=top_k, max_distance=max_distance, mode=mode)
        local = self.corpus.search(q, top_k=top_k, max_distance=max_distance, mode=mode, row_base=self.row_base)
        threshold_all = max_distance is not None and mode == MODE_DOCUMENTS
        out = []
        if threshold_all or max_distance is not None:
            for rows, dd in local:
                out.append(allgather_threshold_hits(rows, dd, None if threshold_all else top_k, self.group))
            return out
        return exchange_topk(local, top_k, ctx=self.corpus.ctx, group=self.group)


class ShardedIvfPq:
    """Row-sharded IVF-PQ: every rank builds an index over ITS rows only (own centroids and codebooks, no
    collective in the build -- ranks are independent, like the exact path), searches it locally with
    row_base, and the per-rank top-k lists go through the same all-gather + merge.  Probing nprobe lists on
    each of R shards reads the same fraction of the codes as nprobe lists of one global index."""

    def __init__(self, index, row_base, group=None):
        self.index = index
        self.row_base = int(row_base)
        self.group = group

    def search(self, queries, top_k=10, nprobe=32, rerank=0):
        local = self.index.search(queries, top_k=top_k, nprobe=nprobe, rerank=rerank, row_base=self.row_base)
        return exchange_topk(local, top_k, ctx=self.index.corpus.ctx, group=self.group)
