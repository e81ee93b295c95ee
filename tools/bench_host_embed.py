"""End-to-end host embedding throughput (create_document_from_content path): tokenise (host threads) || H2D || K1,
through the C++ host layer with the whitespace-vocab tokenizer.  Reports lines/s for one big synthetic file."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import semtools_amd as smt
from semtools_amd import host
from tests import synth

V = 50000
n_lines = int(os.environ.get("LINES", 2_000_000))
table = synth.table(V, seed=2)
lines = synth.pseudo_prose(20000, vocab_size=V, seed=1)
content = "\n".join(lines[i % len(lines)] for i in range(n_lines)) + "\n"
ctx = smt.Context(0)
m = host.StaticModel(ctx, table=table, tokenizer="hash")
for rep in range(2):
    t0 = time.perf_counter()
    out = host.search_content(m, lines[17], content, n_lines=0, top_k=3)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(lines=n_lines, bytes=len(content), seconds=round(dt, 3), lines_per_s=round(n_lines / dt / 1e6, 2),
                          first_hit=out.split("\n")[0])), flush=True)
