"""CPU, world_size 2, gloo: the N>1 path's exchange step.  Each rank owns a contiguous row shard,
produces its sorted top-k list, all-gathers the fixed-size lists and merges redundantly
(semtools_amd/dist.py).  No GPU here, so the per-shard list comes from the oracle (the checker
standing in for the local scan); what is under test is sharding + all-gather + merge: the
result must equal the single-shard answer on every rank."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, k, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from semtools_amd import dist as sdist

    emb = synth.unit_rows(n_rows, seed=3)
    qs = synth.unit_query(4, nq=2)
    b, e = sdist.shard_bounds(n_rows, world)[rank]
    rows = np.full((2, k), -1, np.int64)
    dd = np.full((2, k), np.inf)
    for qi in range(2):
        res = orc.search_documents(emb[b:e], [e - b], qs[qi], 0, k, accurate=True)
        rows[qi, : len(res)] = [r["match_line"] + b for r in res]
        dd[qi, : len(res)] = [r["distance"] for r in res]
    mr, md = sdist.allgather_merge_topk(torch.from_numpy(rows), torch.from_numpy(dd), k)
    # the one-collective packed form must give the same answer
    packed = torch.from_numpy(np.stack([rows, dd.view(np.int64)], axis=1).copy())
    mp = sdist.allgather_merge_packed(packed, k)
    assert torch.equal(mp[:, 0], mr) and torch.equal(mp[:, 1], md.view(torch.int64))
    out_q.put((rank, mr.numpy().tolist(), md.numpy().tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_search_equals_single_shard():
    from oracle import oracle as orc

    n_rows, k, world = 3001, 7, 2      # odd size: shards 1501 + 1500
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    emb = synth.unit_rows(n_rows, seed=3)
    qs = synth.unit_query(4, nq=2)
    for rank, rows, dd in got:
        for qi in range(2):
            ref = orc.search_documents(emb, [n_rows], qs[qi], 0, k, accurate=True)
            assert rows[qi] == [r["match_line"] for r in ref], rank
            assert dd[qi] == [r["distance"] for r in ref], rank


def test_shard_bounds_cover_and_are_contiguous():
    from semtools_amd import dist as sdist

    for n in (0, 1, 7, 8, 9, 100_000_001):
        for w in (1, 2, 3, 8):
            b = sdist.shard_bounds(n, w)
            assert len(b) == w and b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert all(0 <= e - s <= -(-n // w) for s, e in b)
