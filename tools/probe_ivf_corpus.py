"""Which synthetic corpus makes PROBING matter?  (VERDICT r5 missing 3: on the bench's 20 000-topic corpus recall@10 is the same at
nprobe 1 and 128.)  For a few (topics, latent dims, spread) settings of tests/synth's generative model: recall@10 of the shipped
coding at nprobe 1 / 8 / 32 / 128 over --rows rows, nlist 4096, 1000 independent queries.  Run on the GPU box:
python tools/probe_ivf_corpus.py --rows 10000000 > gpurun_out/probe_ivf_corpus.jsonl"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semtools_amd as smt
from tests import synth

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--nlist", type=int, default=4096)
ap.add_argument("--nq", type=int, default=1000)
args = ap.parse_args()
dev = torch.device("cuda:0")
ctx = smt.Context(0)
for topics, latent, spread, noise in ((500, 8, 0.35, 0.01), (500, 32, 0.6, 0.01), (64, 32, 0.8, 0.01), (64, 64, 1.0, 0.02), (8, 64, 1.5, 0.02),
                                      (2000, 24, 0.7, 0.02)):
    gen = synth.clustered_model_torch(topics, latent, 21, dev)
    x = synth.clustered_sample_torch(gen, args.rows, 22, spread=spread, noise=noise)
    q = synth.clustered_sample_torch(gen, args.nq, 23, spread=spread, noise=noise).cpu().numpy()
    del gen
    torch.cuda.synchronize()
    c = smt.Corpus(ctx, device_ptr=x.data_ptr(), rows=args.rows)
    exact = c.search(q, top_k=10)
    d10 = float(np.mean([e[1][-1] for e in exact]))
    row = dict(topics=topics, latent=latent, spread=spread, noise=noise, mean_10th_distance=round(d10, 4))
    for lp in (True, False):
        ix = smt.IvfPq(c, nlist=args.nlist, train_iters=10, local_pca=lp)
        for nprobe in (1, 8, 32, 128):
            got = ix.search(q, top_k=10, nprobe=nprobe, rerank=128)
            hit = sum(len(set(r.tolist()) & set(e.tolist())) for (r, _), (e, _) in zip(got, exact))
            row[("lpca" if lp else "pq") + f"_np{nprobe}"] = round(hit / (args.nq * 10), 4)
        ix.close()
    print(json.dumps(row), flush=True)
    c.close()
    del x
    torch.cuda.empty_cache()
