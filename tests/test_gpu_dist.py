"""GPU: the N>1 glue (semtools_amd/dist.py) on the RCCL backend with the only world size a 1-GPU box offers.
A one-rank "nccl" group still drives the device all-gather buffer layout and the device merge kernel, and
a row_base != 0 shard checks that global row ids come out of the exchange.  The 2-rank exchange logic
itself is covered on CPU (tests/test_dist_cpu.py, gloo); the 8-GPU run is the driver's."""
import os
import socket

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group():
    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        yield None
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    yield None
    dist.destroy_process_group()


def test_sharded_corpus_on_rccl_backend_matches_plain_search(gpu_ctx, nccl_group):
    import semtools_amd as smt
    from semtools_amd import dist as sdist

    emb = synth.unit_rows(6000, seed=3)
    qs = synth.unit_query(4, nq=3)
    c = smt.Corpus(gpu_ctx)
    c.append(emb)
    base = 1_000_000_007
    sc = sdist.ShardedCorpus(c, row_base=base)
    for kwargs in (dict(top_k=7), dict(top_k=7, max_distance=0.93), dict(top_k=4, max_distance=0.95, mode=smt.MODE_WORKSPACE)):
        got = sc.search(qs, **kwargs)
        want = c.search(qs, **kwargs)
        for (gr, gd), (wr, wd) in zip(got, want):
            assert gr.tolist() == (wr + np.uint64(base)).tolist()
            assert np.array_equal(gd, wd)
    c.close()


def test_packed_allgather_merge_on_device(gpu_ctx, nccl_group):
    """allgather_merge_packed on CUDA tensors == host merge of the same lists."""
    import torch
    import semtools_amd as smt
    from semtools_amd import dist as sdist

    rng = np.random.default_rng(11)
    nq, k = 5, 10
    dd = np.sort(rng.random((nq, k)), axis=1)
    rows = rng.integers(0, 1 << 40, (nq, k)).astype(np.uint64)
    ctx = smt.Context(0, stream=torch.cuda.current_stream().cuda_stream)   # merge runs on torch's stream
    packed = torch.from_numpy(np.stack([rows.view(np.int64), dd.view(np.int64)], axis=1).copy()).cuda()
    out = sdist.allgather_merge_packed(packed, 6, ctx=ctx)
    torch.cuda.synchronize()
    m = out.cpu().numpy()
    mr, md, _ = smt.merge_topk(rows[None], dd[None], 6)
    assert np.array_equal(np.ascontiguousarray(m[:, 0]).view(np.uint64), mr)
    assert np.array_equal(np.ascontiguousarray(m[:, 1]).view(np.float64), md)
    ctx.close()


def test_sharded_ivfpq_exchange(gpu_ctx, nccl_group):
    import semtools_amd as smt
    from semtools_amd import dist as sdist

    x = synth.clustered_rows_torch(20000, 64, 8, 2, "cuda").cpu().numpy()
    c = smt.Corpus(gpu_ctx)
    c.append(x)
    ix = smt.IvfPq(c, nlist=64, train_iters=4)
    base = 5_000_000_000
    got = sdist.ShardedIvfPq(ix, row_base=base).search(x[:4], top_k=5, nprobe=8)
    want = ix.search(x[:4], top_k=5, nprobe=8, row_base=base)
    for (gr, gd), (wr, wd) in zip(got, want):
        assert gr.tolist() == wr.tolist() and np.array_equal(gd, wd) and int(gr[0]) >= base
    ix.close()
    c.close()
