"""GPU, BASELINE.json's full sizes (c2: 1 query x 1M rows; c3: batched queries x 10M rows), checked through
size-independent properties instead of the (too slow at this size) CPU oracle:
  * planted rows: copies of the query come back first, in ascending row order, distance exactly 0;
  * agreement with an independent fp64 torch evaluation of the same scores on the GPU;
  * row-sharding + merge == single shard;  batched path == single-query path;  determinism."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make(rows, seed):
    import torch

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.randn(rows, 256, device=dev, generator=g)
    x /= x.norm(dim=1, keepdim=True)
    return x


def test_c2_one_query_one_million_rows(gpu_ctx):
    import torch
    import semtools_amd as smt

    rows = 1_000_000
    x = _make(rows, 3)
    q = _make(1, 4)[0]
    planted = [7, 123_456, 999_999]
    x[planted] = q
    x[[11, 500_000]] = 0.0
    near = q + 0.02 * _make(1, 5)[0]
    x[[42, 77_777]] = near                                  # exact tie pair -> ascending rows
    torch.cuda.synchronize()
    c = smt.Corpus(gpu_ctx, device_ptr=x.data_ptr(), rows=rows)
    qh = q.cpu().numpy()
    r, d = c.search(qh, top_k=10)[0]
    assert r[:5].tolist() == planted + [42, 77_777]
    assert d[:3].tolist() == [0.0, 0.0, 0.0] or (d[:3] < 2.3e-16).all()
    assert d[3] == d[4] and (np.diff(d) >= 0).all()
    ref = 1.0 - (x.double() @ q.double()) / (x.double().norm(dim=1).clamp_min(1e-300) * q.double().norm())
    ref[[11, 500_000]] = 1.0
    tv, ti = torch.topk(ref, 10, largest=False)
    np.testing.assert_allclose(d, np.sort(tv.cpu().numpy()), rtol=0, atol=1e-9)
    assert set(r[5:].tolist()) == set(ti.cpu().tolist()) - set(r[:5].tolist())
    # determinism + threshold mode consistency at full size
    r2, d2 = c.search(qh, top_k=10)[0]
    assert r2.tolist() == r.tolist() and np.array_equal(d2, d)
    rt, dt = c.search(qh, max_distance=float(d[9]) + 1e-12)[0]
    assert rt.tolist() == r.tolist() and np.array_equal(dt, d)
    # 4 contiguous shards + merge == single shard
    lists_r, lists_d = [], []
    for b, e in ((0, 250_000), (250_000, 500_001), (500_001, 999_000), (999_000, rows)):
        s = smt.Corpus(gpu_ctx, device_ptr=x[b:e].data_ptr(), rows=e - b)
        rr, dd = s.search(qh, top_k=10, row_base=b)[0]
        lists_r.append(rr[None]); lists_d.append(dd[None]); s.close()
    mr, md, _ = smt.merge_topk(np.stack(lists_r), np.stack(lists_d), 10)
    assert mr[0].tolist() == r.tolist() and np.array_equal(md[0], d)
    c.close()


def test_c3_batched_queries_ten_million_rows(gpu_ctx):
    """BASELINE config c3 at its full size through the HOST form (queries in, hits out): 1000 queries x 10 M rows.  Every query's
    answer agrees with the single-query scan path (K2: a different kernel, f32 scan + the same exact re-scoring) -- rows and f64
    distances identical, the check bench.py's c3 leg makes -- and a sample with an independent fp64 evaluation.  (The
    device-resident form of the same batch, with the operand image and the which-kernel-answered asserts, is
    tests/test_gpu_defaults.py::test_c3_thousand_queries_ten_million_rows_in_the_default_mode.)"""
    import torch
    import semtools_amd as smt

    rows, nq, k = 10_000_000, 1000, 10
    x = _make(rows, 3)
    q = _make(nq, 5)
    x[1_234_567] = q[0]
    x[[9_999_999, 5]] = q[1]
    torch.cuda.synchronize()
    c = smt.Corpus(gpu_ctx, device_ptr=x.data_ptr(), rows=rows)
    qh = q.cpu().numpy()
    gpu_ctx.uncertain_count()
    got = c.search(qh, top_k=k)                              # K3 path (nq >= 8): f16 x 1 nomination at this batch size
    assert got[0][0][0] == 1_234_567 and got[0][1][0] < 2.3e-16
    assert got[1][0][:2].tolist() == [5, 9_999_999]
    gpu_ctx.set_tuning("gemm_min_nq", 8)                     # (keep 4-query calls on the scan kernel: they are the truth here)
    try:
        n_same = 0
        for i in range(0, nq, 4):
            four = c.search(qh[i:i + 4], top_k=k)            # K2 path, four queries per pass
            n_same += sum(four[j][0].tolist() == got[i + j][0].tolist() and np.array_equal(four[j][1], got[i + j][1]) for j in range(4))
    finally:
        gpu_ctx.set_tuning("gemm_min_nq", 5)
    assert n_same == nq, f"{n_same}/{nq} queries agree with the scan path"
    xn = x.double().norm(dim=1)
    for i in (0, 1, 2, 50, 95, 511, 999):                    # independent fp64 check
        ref = 1.0 - (x.double() @ q[i].double()) / (xn * q[i].double().norm())
        tv, ti = torch.topk(ref.clamp_min(0.0), k, largest=False)
        assert got[i][0].tolist() == ti.cpu().tolist(), i
        np.testing.assert_allclose(got[i][1], tv.cpu().numpy(), rtol=0, atol=1e-9)
    c.close()


def test_threshold_mode_with_more_than_a_million_hits(gpu_ctx):
    """Unbounded-result mode at scale (mod.rs:115-116 returns ALL hits): the hit buffer starts at 2^20 rows and
    must be re-sized when every one of 1.2M rows is under the threshold."""
    import torch
    import semtools_amd as smt

    rows = 1_200_000
    x = _make(rows, 9)
    q = _make(1, 10)[0]
    x[[5, 600_000]] = q
    torch.cuda.synchronize()
    c = smt.Corpus(gpu_ctx, device_ptr=x.data_ptr(), rows=rows)
    r, d = c.search(q.cpu().numpy(), top_k=3, max_distance=2.5)[0]
    assert r.size == rows and r[:2].tolist() == [5, 600_000] and d[0] < 1e-15
    assert (np.diff(d) >= 0).all() and len(np.unique(r)) == rows
    ties = np.nonzero(np.diff(d) == 0)[0]
    assert (r[ties] < r[ties + 1]).all()                    # equal distances come back in row order
    r10, d10 = c.search(q.cpu().numpy(), top_k=10)[0]
    assert r[:10].tolist() == r10.tolist() and np.array_equal(d[:10], d10)
    c.close()


def test_c2_spec_corpus_duplicates_zero_rows_k_sweep_and_threshold(gpu_ctx):
    """SURVEY 8(d) config c2 as specified: 1 M unit rows, 1 % exact-duplicate rows, 0.1 % all-zero rows,
    k in {3, 10, 100}, and the threshold run max_distance = 0.9 -- at a size the CPU oracle does not finish
    in seconds, so checked through properties: agreement with an independent fp64 evaluation on the GPU,
    (distance, row) order with exact ties among duplicates, count + index checksum of the threshold hits."""
    import torch
    import semtools_amd as smt

    rows = 1_000_000
    dev = torch.device("cuda:0")
    x = _make(rows, 3)
    q = _make(1, 4)[0]
    g = torch.Generator(device=dev)
    g.manual_seed(33)
    perm = torch.randperm(rows, generator=g, device=dev)
    dup_dst, zero = perm[:10_000], perm[10_000:11_000]
    x[dup_dst] = x[torch.randint(0, rows, (10_000,), generator=g, device=dev)]
    best = torch.topk(x @ q, 3).indices                         # make sure duplicates sit INSIDE the top-k
    x[perm[11_000:11_006]] = x[best.repeat(2)]
    x[zero] = 0.0
    torch.cuda.synchronize()
    xd = x.double()
    ref = 1.0 - (xd @ q.double()) / (xd.norm(dim=1).clamp_min(1e-300) * q.double().norm())
    ref[zero] = 1.0                                             # simsimd: ab == 0 -> distance 1
    c = smt.Corpus(gpu_ctx, device_ptr=x.data_ptr(), rows=rows)
    qh = q.cpu().numpy()

    for k in (3, 10, 100):                                      # 100 > 64 goes through the large-k path
        r, d = c.search(qh, top_k=k)[0]
        assert len(r) == k
        tv, _ = torch.topk(ref, k, largest=False)
        np.testing.assert_allclose(d, np.sort(tv.cpu().numpy()), rtol=0, atol=1e-9)
        assert all((d[i], r[i]) < (d[i + 1], r[i + 1]) for i in range(k - 1))      # distance asc, row asc on ties
        must = torch.nonzero(ref < tv[-1] - 1e-9).flatten().cpu().tolist()
        assert set(must) <= set(r.tolist())
        if k >= 10:                                             # 3 best rows x 3 copies each: exact ties, ascending rows
            assert d[0] == d[1] == d[2] and d[3] == d[4] == d[5] and d[6] == d[7] == d[8]
            assert sorted(r[:9].tolist()) == sorted(best.tolist() + perm[11_000:11_006].tolist())

    rt, dt = c.search(qh, max_distance=0.9)[0]
    inside = ref < 0.9
    near = int(((ref - 0.9).abs() < 1e-9).sum())
    assert abs(len(rt) - int(inside.sum())) <= near and len(rt) > 10_000
    assert (dt < 0.9).all() and all(np.diff(dt) >= 0)
    tie = np.nonzero(np.diff(dt) == 0)[0]
    assert len(tie) > 0 and (np.diff(rt.astype(np.int64))[tie] > 0).all()          # duplicates: ties in row order
    if near == 0:
        assert int(rt.astype(np.int64).sum()) == int(torch.nonzero(inside).sum())
        assert not bool(inside[zero].any()) and not (set(zero.cpu().tolist()) & set(rt.tolist()))
    c.close()


def test_ivf_index_encodes_every_row_of_a_seventy_million_row_shard(gpu_ctx):
    """Regression (round 3): the per-list PCA encoder launched rows x 64 work-items -- more than a dispatch can carry
    (2^32 - 1) from 67 M rows on -- and at config c5's 100 M rows silently left two thirds of the codes unwritten
    (recall 0.35 whatever nprobe and re-score depth).  70 M rows: a query planted in the LAST million rows must be found."""
    import torch
    import semtools_amd as smt

    rows = 70_000_000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(21)
    centres = torch.randn(512, 256, device=dev, generator=g)
    x = torch.empty((rows, 256), device=dev)
    for b in range(0, rows, 2_000_000):
        e = min(rows, b + 2_000_000)
        c = centres[torch.randint(0, 512, (e - b,), device=dev, generator=g)] + 0.35 * torch.randn(e - b, 256, device=dev, generator=g)
        x[b:e] = c / c.norm(dim=1, keepdim=True)
    probes = torch.tensor([5, 20_000_000, 44_000_000, 67_200_000, 69_000_000, rows - 1], device=dev)
    q = x[probes] + 0.01 * torch.randn(len(probes), 256, device=dev, generator=g)
    torch.cuda.synchronize()
    c = smt.Corpus(gpu_ctx, device_ptr=x.data_ptr(), rows=rows)
    ix = smt.IvfPq(c, nlist=1024, train_iters=4, local_pca=True)
    got = ix.search(q.cpu().numpy(), top_k=5, nprobe=4, rerank=64)
    exact = c.search(q.cpu().numpy(), top_k=5)
    for i, p in enumerate(probes.tolist()):
        assert got[i][0][0] == p == exact[i][0][0], (i, got[i][0][:3], exact[i][0][:3])     # (codes of the last rows exist)
    ix.close()
    c.close()
