"""Multi-GPU glue for Python hosts: one process per GPU (torchrun), corpus row-sharded, ONE exchange step.

The exchange lives INSIDE libsemtools_hip.so (csrc/group.cpp: smt_group_* / smt_sharded_*: per-shard scan -> ncclAllGather of the
packed k-lists -> merge_topk_kernel); this module only joins the ranks of a torch.distributed job into one library group (rank 0's
ncclUniqueId travels through torch's store).  core.ShardedCorpus / core.ShardedIvfPq / core.ShardedModel hand the work to that
group.  (The torch-tensor restatement of the exchange that rounds 1-3 kept here for the gloo tests is tests/dist_protocol.py now.)
"""
import torch.distributed as dist

from . import core


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
