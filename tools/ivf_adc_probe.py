"""Workload for the counters of ivf_adc_kernel: the c5 leg of bench.py (10 M rows in 20000 topics, nlist 4096, nprobe 8, 128 ADC
candidates per list re-scored, 1000 queries), 5 device-resident batches per coding.  Run under rocprofv3 --pmc by tools/gpu_r04d.sh."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt  # noqa: E402
from tests import synth  # noqa: E402

rows, nq, k = 10_000_000, 1000, 10
dev = torch.device("cuda:0")
gen = synth.clustered_model_torch(20000, 8, 11, dev)
x = synth.clustered_sample_torch(gen, rows, 12)
qd = synth.clustered_sample_torch(gen, nq, 13)
del gen
torch.cuda.synchronize()
ctx = smt.Context(0)
corpus = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=rows)
o_r = torch.empty((nq, k), dtype=torch.int64, device=dev)
o_d = torch.empty((nq, k), dtype=torch.float64, device=dev)
out = {}
for local_pca in (True, False):
    ix = smt.IvfPq(corpus, nlist=4096, train_iters=10, local_pca=local_pca)
    ix.search_device(qd.data_ptr(), nq, k, 8, 128, 0, o_r.data_ptr(), o_d.data_ptr())
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ix.search_device(qd.data_ptr(), nq, k, 8, 128, 0, o_r.data_ptr(), o_d.data_ptr())
    ctx.synchronize()
    out["lpca" if local_pca else "pq"] = {"ms_per_batch": (time.perf_counter() - t0) / 5 * 1e3}
    ix.close()
print(json.dumps(out))
