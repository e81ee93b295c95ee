"""CPU, world_size 2, gloo: the N>1 path's exchange step.  Each rank owns a contiguous row shard,
produces its sorted top-k list, all-gathers the fixed-size lists and merges redundantly
(tests/dist_protocol.py: the library's exchange protocol restated over CPU tensors).  No GPU here, so the per-shard list comes from the oracle (the checker
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
    from tests import dist_protocol as sdist

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
    from tests import dist_protocol as sdist

    for n in (0, 1, 7, 8, 9, 100_000_001):
        for w in (1, 2, 3, 8):
            b = sdist.shard_bounds(n, w)
            assert len(b) == w and b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert all(0 <= e - s <= -(-n // w) for s, e in b)


class _OracleShard:
    """Duck-typed stand-in for semtools_amd.core.Corpus on a machine without a GPU: the local scan is the
    oracle (checker standing in for the shard's smt_search).  Only the exchange logic is under test."""

    ctx = None

    def __init__(self, emb):
        self.emb = emb

    def search(self, q, top_k, max_distance, mode, row_base):
        from oracle import oracle as orc

        out = []
        for qi in range(q.shape[0]):
            if mode == 0:
                res = orc.search_documents(self.emb, [len(self.emb)], q[qi], 0, top_k, max_distance=max_distance,
                                           accurate=True)
                rows = [r["match_line"] + row_base for r in res]
                dd = [r["distance"] for r in res]
            else:
                n = len(self.emb)
                res = orc.search_line_embeddings(self.emb, np.zeros(n, np.uint32), np.arange(n, dtype=np.int32), q[qi],
                                                 [0], top_k, max_distance)
                rows = [r["row"] + row_base for r in res]
                dd = [float(r["distance"]) for r in res]
            out.append((np.asarray(rows, np.uint64), np.asarray(dd, np.float64)))
        return out


def _worker_sharded(rank, world, port, n_rows, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import dist_protocol as sdist

    emb = synth.unit_rows(n_rows, seed=3)
    qs = synth.unit_query(4, nq=2)
    b, e = sdist.shard_bounds(n_rows, world)[rank]
    sc = sdist.ShardedCorpus(_OracleShard(emb[b:e]), row_base=b)
    res = {
        "topk": sc.search(qs, 5),
        "thr_all": sc.search(qs, 5, max_distance=0.93),                 # A6: all hits, top_k ignored
        "thr_ws": sc.search(qs, 4, max_distance=0.95, mode=1),          # A10: threshold, then top_k
        "thr_none": sc.search(qs, 5, max_distance=1e-9),                # nobody has a hit
    }
    out_q.put((rank, {k: [(r.tolist(), d.tolist()) for r, d in v] for k, v in res.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_corpus_all_modes_equal_single_shard():
    n_rows, world = 2501, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded, args=(r, world, port, n_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    emb = synth.unit_rows(n_rows, seed=3)
    qs = synth.unit_query(4, nq=2)
    whole = _OracleShard(emb)
    want = {
        "topk": whole.search(qs, 5, None, 0, 0),
        "thr_all": whole.search(qs, 5, 0.93, 0, 0),
        "thr_ws": whole.search(qs, 4, 0.95, 1, 0),
        "thr_none": whole.search(qs, 5, 1e-9, 0, 0),
    }
    assert len(want["thr_all"][0][0]) > 5          # the threshold case really returns more than top_k
    for rank, res in got:
        for key, per_q in want.items():
            for qi, (rows, dd) in enumerate(per_q):
                assert res[key][qi][0] == rows.tolist(), (rank, key, qi)
                assert res[key][qi][1] == dd.tolist(), (rank, key, qi)
