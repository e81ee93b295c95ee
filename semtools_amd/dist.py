"""Multi-GPU glue: one process per GPU, corpus row-sharded, ONE exchange step.

Each rank scans its own contiguous row range and produces a sorted top-k list
(distance asc, global row asc).  The only collective on the path is an
all-gather of those fixed-size lists (k x 16 B per query per rank: latency
bound, xGMI bandwidth irrelevant), after which every rank merges redundantly.
With backend "nccl" the all-gather is RCCL over xGMI on device buffers; the
same code runs on "gloo"/CPU tensors, which is how it is tested without GPUs.
There is no reference counterpart (the reference is single-process CPU code);
the contract is: sharded result == single-shard result (tests/).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import core

PAD_ROW = -1  # UINT64_MAX viewed as int64


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
