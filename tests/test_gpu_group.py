"""GPU: groups of GPUs behind the C ABI (include/semtools_hip.h: smt_group_* / smt_sharded_*).

A 1-GPU box offers two things: (i) the real RCCL path with ONE rank -- ncclCommInitAll / ncclCommInitRank,
ncclAllGather, device merge -- and (ii) "logical" groups, N ranks on one device (RCCL refuses two ranks on one GPU) whose
k-lists meet through the PEER transport -- the merge kernel reads every rank's list in place, one event per rank, exactly
what a one-process group of N real GPUs does over xGMI -- or, as the fallback, through event-ordered device copies.  (ii)
drives everything except the RCCL call itself with N > 1: HIP scan on every shard -> exchange -> merge kernel, range
localisation, global row numbering, the variable-length threshold exchange.
Contract everywhere: sharded result == smt_search on the unsharded matrix (which the other GPU tests pin to the
oracle)."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _same(got, want):
    assert len(got) == len(want)
    for (gr, gd), (wr, wd) in zip(got, want):
        assert gr.tolist() == wr.tolist()
        assert np.array_equal(gd, wd)


CASES = [
    dict(top_k=7),
    dict(top_k=56),
    dict(top_k=7, max_distance=0.93),                                   # all rows under the threshold (A6)
    dict(top_k=4, max_distance=0.95, mode=1),                           # workspace: score threshold, then top-k (A10)
    dict(top_k=100),                                                    # beyond the scan path: all-keys path per shard
    dict(top_k=5, ranges=[(10, 50), (2999, 3001), (4000, 5999)]),       # path-subset filter crossing shard borders
    dict(top_k=3, ranges=[(5990, 6000)]),                               # a filter that leaves most shards nothing
    dict(top_k=9, max_distance=0.9, ranges=[(100, 4100)]),
]


@pytest.fixture(scope="module")
def plain(gpu_ctx):
    import semtools_amd as smt

    emb = synth.unit_rows(6000, seed=3)
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    yield emb, c
    c.close()


def test_rccl_group_of_one_rank(plain):
    """smt_group_create -> ncclCommInitAll on the one GPU there is; the communicator itself reports its size."""
    import semtools_amd as smt

    emb, c = plain
    g = smt.Group([0])
    info = g.info()
    assert info["n_ranks"] == 1 and info["n_local"] == 1 and info["rccl_ranks"] == 1 and info["rccl_version"] > 20000
    assert g.transport == "peer"                      # one-process group: lists are read in place by default
    with pytest.raises(RuntimeError):
        g.set_transport("copy")                       # (the all-gather of logical groups; this one has a communicator)
    sc = smt.ShardedCorpus(g, rows=emb)
    # the device form through both transports of a real group: peer reads and ncclAllGather
    import torch
    k, nq = 10, 3
    qd = torch.from_numpy(synth.unit_query(8, nq=nq)).cuda()
    want = c.search(qd.cpu().numpy(), top_k=k)
    for transport in ("rccl", "peer"):
        g.set_transport(transport)
        assert g.transport == transport
        out = torch.zeros((nq, 2, k), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        sc.search_topk_device([qd.data_ptr()], nq, k, [out.data_ptr()])
        g.synchronize()
        m = out.cpu().numpy()
        for i in range(nq):
            assert np.ascontiguousarray(m[i, 0]).view(np.uint64).tolist() == want[i][0].tolist()
            assert np.array_equal(np.ascontiguousarray(m[i, 1]).view(np.float64), want[i][1])
    assert sc.rows == len(emb) and sc.rank_rows().tolist() == [len(emb)]
    qs = synth.unit_query(4, nq=3)
    for kw in CASES:
        _same(sc.search(qs, **kw), c.search(qs, **kw))
    qb = synth.unit_query(5, nq=12)                      # >= 8 queries: the MFMA path inside the shard
    _same(sc.search(qb, top_k=10), c.search(qb, top_k=10))
    sc.close()
    g.close()


def test_rank_per_process_group_of_one(plain):
    """The torchrun form: unique id -> ncclCommInitRank."""
    import semtools_amd as smt

    emb, c = plain
    uid = smt.Group.unique_id()
    assert len(uid) == 128
    g = smt.Group.from_rank(device=0, rank=0, n_ranks=1, unique_id=uid)
    assert g.info()["rccl_ranks"] == 1
    sc = smt.ShardedCorpus(g, rows=emb)
    qs = synth.unit_query(6, nq=2)
    _same(sc.search(qs, top_k=5), c.search(qs, top_k=5))
    sc.close()
    g.close()


@pytest.mark.parametrize("transport", ["peer", "copy"])
@pytest.mark.parametrize("n_shards", [2, 3, 8])
def test_logical_shards_equal_the_unsharded_search(plain, n_shards, transport):
    import semtools_amd as smt

    emb, c = plain
    g = smt.Group.logical(0, n_shards)
    assert g.info()["n_ranks"] == n_shards and g.info()["rccl_ranks"] == 0
    assert g.transport == "peer"
    with pytest.raises(RuntimeError):
        g.set_transport("rccl")                       # no communicator behind logical ranks
    g.set_transport(transport)
    assert g.transport == transport
    sc = smt.ShardedCorpus(g, rows=emb)
    per = -(-len(emb) // n_shards)
    assert sc.rank_rows().tolist() == [max(0, min(per, len(emb) - r * per)) for r in range(n_shards)]
    qs = synth.unit_query(4, nq=3)
    for kw in CASES:
        _same(sc.search(qs, **kw), c.search(qs, **kw))
    qb = synth.unit_query(5, nq=12)
    _same(sc.search(qb, top_k=10), c.search(qb, top_k=10))
    ql = synth.unit_query(6, nq=140)                     # 128+ queries: the f16 x 2 nomination inside every shard, lists merged
    _same(sc.search(ql, top_k=7), c.search(ql, top_k=7))
    _same(sc.search(ql[:5], max_distance=0.88), c.search(ql[:5], max_distance=0.88))   # several threshold queries per shard
    # rows appended later extend the last rank's range
    extra = synth.unit_rows(700, seed=77)
    first = sc.append(extra)
    assert first == len(emb) and sc.rows == len(emb) + 700
    c2 = smt.Corpus(c.ctx)
    c2.append(np.concatenate([emb, extra]))
    _same(sc.search(qs, top_k=12), c2.search(qs, top_k=12))
    c2.close()
    sc.close()
    g.close()


@pytest.mark.parametrize("n_shards", [2, 5])
def test_variable_length_lists_through_the_exchange_of_a_multi_process_group(plain, n_shards, monkeypatch):
    """Threshold mode and k > 56 produce per-rank host lists.  A one-process group merges them where they are; ranks in different
    processes send them through a count all-gather and one padded all-gather.  $SEMTOOLS_GROUP_HOST_LISTS=exchange runs THAT path
    on logical ranks (the only way to run it on one GPU): same answers as the unsharded search, with and without it."""
    import semtools_amd as smt

    emb, c = plain
    qs = synth.unit_query(8, nq=5)
    cases = [dict(top_k=3, max_distance=0.9), dict(top_k=3, max_distance=0.2), dict(top_k=100), dict(top_k=64), dict(top_k=2000),
             dict(top_k=5, max_distance=0.93, ranges=[(10, 900), (2000, 2600)])]
    for how in ("exchange", "direct"):
        if how == "exchange":
            monkeypatch.setenv("SEMTOOLS_GROUP_HOST_LISTS", "exchange")
        else:
            monkeypatch.delenv("SEMTOOLS_GROUP_HOST_LISTS", raising=False)
        g = smt.Group.logical(0, n_shards)
        sc = smt.ShardedCorpus(g, rows=emb)
        try:
            for kw in cases:
                _same(sc.search(qs, **kw), c.search(qs, **kw))
        finally:
            sc.close()
            g.close()


@pytest.mark.parametrize("transport", ["peer", "copy"])
def test_sharded_device_form_and_file_round_trip(plain, tmp_path, transport):
    """smt_sharded_search_topk_device (what bench.py --gpus N times) + save/load through the corpus file."""
    import torch
    import semtools_amd as smt

    emb, c = plain
    g = smt.Group.logical(0, 4)
    g.set_transport(transport)
    sc = smt.ShardedCorpus(g, rows=emb)
    path = tmp_path / "corpus.f32"
    sc.save(path)
    c_file = smt.Corpus.load(c.ctx, path)
    assert np.array_equal(c_file.read_rows(0, len(emb)), emb)
    c_file.close()
    sc2 = smt.ShardedCorpus.load(g, path)
    assert sc2.rank_rows().tolist() == sc.rank_rows().tolist()
    k, nq = 10, 3
    qs = synth.unit_query(8, nq=nq)
    qd = torch.from_numpy(qs).cuda()
    outs = [torch.empty((nq, 2, k), dtype=torch.int64, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    sc2.search_topk_device([qd.data_ptr()] * 4, nq, k, [o.data_ptr() for o in outs])
    g.synchronize()
    want = c.search(qs, top_k=k)
    for o in outs:                                     # every rank holds the same merged answer
        m = o.cpu().numpy()
        for i in range(nq):
            assert np.ascontiguousarray(m[i, 0]).view(np.uint64).tolist() == want[i][0].tolist()
            assert np.array_equal(np.ascontiguousarray(m[i, 1]).view(np.float64), want[i][1])
    # pipelined single-query form (async select + gather + merge on the aux streams), answers land in pinned memory
    for i in range(4):
        g.ctx(i).set_tuning("async_select", 1)
    ring = torch.empty((6, 2, k), dtype=torch.int64).pin_memory()
    for step in range(6):
        q1 = qd[step % nq]
        sc2.search_topk_device([q1.data_ptr()] * 4, 1, k, [ring[step].data_ptr(), 0, 0, 0])
    g.synchronize()
    for step in range(6):
        assert ring[step, 0].numpy().view(np.uint64).tolist() == want[step % nq][0].tolist()
        assert np.array_equal(ring[step, 1].numpy().view(np.float64), want[step % nq][1])
    for i in range(4):
        g.ctx(i).set_tuning("async_select", 0)
        assert g.ctx(i).uncertain_count() == 0
    sc.close(); sc2.close(); g.close()


@pytest.mark.parametrize("transport", ["peer", "copy"])
def test_back_to_back_exchanges_keep_their_answers_apart(plain, transport):
    """Nothing synchronises between the calls: rank j's list of exchange e + 1 may only overwrite the one of exchange e after
    every merge of e has read it, and a merge may only read a list its rank has finished writing (peer transport: ev_done /
    ev_ready).  300 searches back to back, the merging device and the query count changing from call to call, the
    host-in / host-out form (which lays the same buffers out differently) mixed in; every answer is checked."""
    import torch
    import semtools_amd as smt

    emb, c = plain
    n = 5
    g = smt.Group.logical(0, n)
    g.set_transport(transport)
    sc = smt.ShardedCorpus(g, rows=emb)
    k = 10
    qs = synth.unit_query(21, nq=16)
    qd = torch.from_numpy(qs).cuda()
    want = c.search(qs, top_k=k)
    steps = 300
    outs = torch.zeros((steps, 3, 2, k), dtype=torch.int64, device="cuda")
    outs2 = torch.zeros((steps, 3, 2, k), dtype=torch.int64, device="cuda")   # the second copy of an answer two devices asked for
    torch.cuda.synchronize()
    for async_select in (0, 1):
        for i in range(n):
            g.ctx(i).set_tuning("async_select", async_select)
        outs.zero_()
        torch.cuda.synchronize()
        plan = []
        for s in range(steps):
            nq = 1 if s % 3 else 3
            q0 = s % (16 - nq + 1)
            merger = s % n
            if s % 50 == 49:
                got = sc.search(qs[q0:q0 + nq], top_k=k)          # host form in between (synchronises)
                for j in range(nq):
                    assert got[j][0].tolist() == want[q0 + j][0].tolist()
            ptrs = [0] * n
            ptrs[merger] = outs[s].data_ptr()
            if s % 7 == 0:
                ptrs[(merger + 2) % n] = outs2[s].data_ptr()                # two devices want this answer
            sc.search_topk_device([qd[q0].data_ptr()] * n, nq, k, ptrs)
            plan.append((q0, nq))
        g.synchronize()
        m, m2 = outs.cpu().numpy(), outs2.cpu().numpy()
        for s, (q0, nq) in enumerate(plan):
            if s % 7 == 0:
                assert np.array_equal(m2[s, :nq], m[s, :nq]), (transport, async_select, s)
            for j in range(nq):
                assert np.ascontiguousarray(m[s, j, 0]).view(np.uint64).tolist() == want[q0 + j][0].tolist(), (transport, async_select, s)
                assert np.array_equal(np.ascontiguousarray(m[s, j, 1]).view(np.float64), want[q0 + j][1])
    for i in range(n):
        g.ctx(i).set_tuning("async_select", 0)
        assert g.ctx(i).uncertain_count() == 0
    sc.close(); g.close()


def test_adopted_device_shards_of_unequal_size(plain):
    import torch
    import semtools_amd as smt

    emb, c = plain
    g = smt.Group.logical(0, 3)
    cuts = [0, 1000, 1000, 6000]                       # the middle shard is empty
    parts = [torch.from_numpy(emb[cuts[i]:cuts[i + 1]].copy()).cuda() for i in range(3)]
    torch.cuda.synchronize()
    sc = smt.ShardedCorpus(g, device_ptrs=[p.data_ptr() if p.numel() else 0 for p in parts],
                           shard_rows=[len(p) for p in parts])
    assert sc.rank_rows().tolist() == [1000, 0, 5000]
    qs = synth.unit_query(4, nq=2)
    for kw in (dict(top_k=6), dict(top_k=6, max_distance=0.92), dict(top_k=3, ranges=[(900, 1100)])):
        _same(sc.search(qs, **kw), c.search(qs, **kw))
    sc.close(); g.close()


@pytest.mark.timeout(180)
def test_sharded_ivf_with_shared_centroids(gpu_ctx):
    """SURVEY 8(e) "C2": data-parallel k-means -- every rank accumulates the centroid sums of ITS sample, the sums are
    all-reduced, every rank finalises the same centroids.  With shared centroids a list means the same thing on every
    shard (per-list sizes of two shards correlate); with independent builds it does not.  Both forms search through
    the same all-gather + merge and return exact (row, distance) pairs."""
    import semtools_amd as smt
    from tests.test_gpu_ivfpq import clustered, recall

    x, _ = clustered(90000, 400, seed=12)
    qs, _ = clustered(48, 400, seed=12)
    plain = smt.Corpus(gpu_ctx)
    plain.append(x)
    exact = plain.search(qs, top_k=10)
    g = smt.Group.logical(0, 3)
    sc = smt.ShardedCorpus(g, rows=x)
    shared = smt.ShardedIvfPq(sc, nlist=128, train_iters=6, shared_centroids=True)
    by_peer = shared.search(qs, top_k=10, nprobe=16, rerank=128)
    g.set_transport("copy")
    by_copy = shared.search(qs, top_k=10, nprobe=16, rerank=128)
    g.set_transport("peer")
    _same(by_peer, by_copy)
    indep = smt.ShardedIvfPq(sc, nlist=128, train_iters=6, shared_centroids=False)
    s0, s1 = shared.shard_list_sizes(0, 128).astype(float), shared.shard_list_sizes(1, 128).astype(float)
    i0, i1 = indep.shard_list_sizes(0, 128).astype(float), indep.shard_list_sizes(1, 128).astype(float)
    assert s0.sum() == 30000 and s1.sum() == 30000
    assert np.corrcoef(s0, s1)[0, 1] > 0.9 > abs(np.corrcoef(i0, i1)[0, 1])
    for ix in (shared, indep):
        got = ix.search(qs, top_k=10, nprobe=16, rerank=128)
        assert recall(got, exact) >= 0.97
        for (rows, dist), q in zip(got, qs):
            assert (np.diff(dist) >= 0).all() and len(set(rows.tolist())) == len(rows)
            ref = 1.0 - x[rows.astype(np.int64)].astype(np.float64) @ q.astype(np.float64)
            assert np.allclose(dist, ref, atol=1e-6)
    # one RCCL rank: ncclAllReduce path of the same build
    g1 = smt.Group([0])
    sc1 = smt.ShardedCorpus(g1, rows=x)
    one = smt.ShardedIvfPq(sc1, nlist=128, train_iters=6, shared_centroids=True)
    assert recall(one.search(qs, top_k=10, nprobe=16, rerank=128), exact) >= 0.97
    for o in (one, shared, indep):
        o.close()
    sc1.close(); g1.close(); sc.close(); g.close(); plain.close()
