"""GPU: handles are not thread-safe, CONTEXTS are independent -- a host that serves several sessions from several threads gives each
its own smt_ctx (the reference is one synchronous thread, src/bin/semtools.rs:134-135; SURVEY 8(b) "Threading").  Four threads, each
with its own context, corpus, model and index on the same GPU, run every search form concurrently (ctypes releases the GIL inside the
calls); every answer must be what the same call returns alone.  Catches process-wide state in the library: static scratch, function
attributes set per context, the thread-local error string."""
import threading

import numpy as np
import pytest

from oracle import oracle as orc
from tests import synth

pytestmark = pytest.mark.gpu


def test_four_contexts_on_four_threads_do_not_see_each_other():
    import torch  # noqa: F401  (binds libamdhip64 first, as conftest does)
    import semtools_amd as smt

    table = synth.table(3000, seed=2)
    n_threads, rounds = 4, 12
    work = []
    for t in range(n_threads):
        emb = synth.unit_rows(20_000 + 3_000 * t, seed=100 + t, dup_frac=0.02, zero_frac=0.002)
        qs = synth.unit_query(200 + t, nq=40)
        ids, offsets = synth.token_lines(500 + 50 * t, V=3000, seed=300 + t, min_tok=0, max_tok=30)
        work.append((emb, qs, ids, offsets))

    def answers(ctx, t, emb, qs, ids, offsets):
        model = smt.Model(ctx, table, normalize=True)
        c = smt.Corpus(ctx)
        c.append(emb)
        out = []
        try:
            for r in range(rounds):
                k = 1 + (r * 7 + t) % 50
                out.append(c.search(qs[:1], top_k=k))                                       # scan kernel
                out.append(c.search(qs[:3], top_k=5))                                       # scan kernel, several queries
                out.append(c.search(qs, top_k=10))                                          # batched kernel
                out.append(c.search(qs[:2], top_k=3, max_distance=0.9))                     # threshold mode
                out.append(c.search(qs[:2], top_k=70))                                      # all-keys path
                out.append(c.search(qs[:9], top_k=4, max_distance=0.95, mode=smt.MODE_WORKSPACE,
                                    ranges=[(10, 5000), (7000, 7001), (9000, 15000)]))      # document subset
                e, _ = model.embed(ids, offsets, max_tokens=16)
                out.append([(np.arange(len(e)), e)])
                with pytest.raises(RuntimeError):                                            # the error string is the thread's own
                    c.search(qs[:1], top_k=3, ranges=[(5, 4)])
        finally:
            c.close()
            model.close()
        return out

    ctxs = [smt.Context(0) for _ in range(n_threads)]
    try:
        alone = [answers(ctxs[t], t, *work[t]) for t in range(n_threads)]
        # the first round's answers against the oracle (the rest of the suite pins every form; here: that `alone` is sane)
        res = orc.search_documents(work[0][0], [len(work[0][0])], work[0][1][0], 0, 1, accurate=True)
        assert alone[0][0][0][0].tolist() == [r["match_line"] for r in res]
        together = [None] * n_threads
        errors = []

        def run(t):
            try:
                together[t] = answers(ctxs[t], t, *work[t])
            except BaseException as exc:   # noqa: BLE001
                errors.append((t, repr(exc)))

        threads = [threading.Thread(target=run, args=(t,)) for t in range(n_threads)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors
        for t in range(n_threads):
            assert len(together[t]) == len(alone[t])
            for call, (a, b) in enumerate(zip(alone[t], together[t])):
                assert len(a) == len(b), (t, call)
                for (ra, da), (rb, db) in zip(a, b):
                    assert np.array_equal(ra, rb) and np.array_equal(da, db), (t, call)
    finally:
        for c in ctxs:
            c.close()
