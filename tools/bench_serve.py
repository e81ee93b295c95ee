"""The batched-query surface end to end: a resident host session (`semtools serve` / smt_host_session_*) over N
lines embedded once, then batches of B queries answered WHOLE -- tokenise + embed the queries, batched search, format the text the
CLI prints.  Queries/s per batch size, and the same batches with the corpus' operand image forbidden (tuning key corpus_image = 0)."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import synth  # noqa: E402


def phases():
    import ctypes as C
    from semtools_amd import _lib as L
    p = L.lib().smt_host_timing_json()
    txt = C.cast(p, C.c_char_p).value.decode()
    L.lib().smt_host_free(C.c_void_p(p))
    return json.loads(txt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vocab", type=int, default=500_000)
    ap.add_argument("--lines", type=int, default=2_000_000)
    ap.add_argument("--files", type=int, default=100)
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 8, 64, 256, 1024])
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from safetensors.numpy import save_file
    import torch  # noqa: F401
    import semtools_amd as smt
    from semtools_amd import host

    tmp = tempfile.mkdtemp(prefix="smt_serve_")
    model_dir = os.path.join(tmp, "model")
    os.makedirs(model_dir)
    V = args.vocab
    rng = np.random.default_rng(2)
    save_file({"embeddings": rng.standard_normal((V, 256), dtype=np.float32) * np.float32(0.1)}, os.path.join(model_dir, "model.safetensors"))
    with open(os.path.join(model_dir, "vocab.txt"), "w") as f:
        f.write("".join(f"w{i}\n" for i in range(V - 1)) + "[UNK]\n")
    json.dump({"normalize": True, "unk_token": "[UNK]"}, open(os.path.join(model_dir, "config.json"), "w"))
    per = args.lines // args.files
    pool = synth.pseudo_prose(per * args.files, vocab_size=V - 1, seed=1)     # every line distinct (a pool of repeated lines makes
    files = []                                                                # every answer a 40-way tie: the exhaustive path)
    for i in range(args.files):
        p = os.path.join(tmp, f"doc{i:03d}.txt")
        with open(p, "w") as f:
            f.write("\n".join(pool[i * per:(i + 1) * per]) + "\n")
        files.append(p)
    queries = synth.pseudo_prose(max(args.batches), vocab_size=V - 1, seed=9)
    result = {"lines": per * args.files, "model": f"synthetic V={V} x 256 f32 + vocab tokenizer", "top_k": 3, "context_lines": 1, "runs": {}}
    for image in (1, 0):
        ctx = smt.Context(0)
        ctx.set_tuning("corpus_image", image)
        m = host.StaticModel(ctx, model_dir=model_dir)
        t0 = time.perf_counter()
        s = host.Session(m, files)
        open_s = time.perf_counter() - t0
        run = {"open_s": round(open_s, 3), "batches": {}}
        texts = {}
        for b in args.batches:
            s.search(queries[:b], n_lines=1, top_k=3)          # warm-up (the first batch of >= 8 builds the image)
            reps = max(3, min(50, 2000 // b))
            ph0 = phases()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = s.search(queries[:b], n_lines=1, top_k=3)
            dt = (time.perf_counter() - t0) / reps
            texts[b] = out
            ph1 = phases()
            run["batches"][b] = {"ms_per_batch": round(dt * 1e3, 3), "queries_per_s": round(b / dt),
                                 "phases_ms_per_batch": {k: round((ph1[k] - ph0.get(k, 0.0)) / reps, 3) for k in ph1 if k.startswith("session_")}}
        result["runs"]["operand_image" if image else "f32_rows_only"] = run
        result.setdefault("_texts", {})[image] = texts
        s.close()
        m.close()
    t = result.pop("_texts")
    result["answers_identical"] = all(t[1][b] == t[0][b] for b in args.batches)
    # ---- the C++ caller: `semtools serve --batch 1024` fed 65 536 queries on stdin, phases from SEMTOOLS_TIMING
    import subprocess
    cli = os.path.join(ROOT, "semtools_amd", "bin", "semtools")
    many = synth.pseudo_prose(65536, vocab_size=V - 1, seed=10)
    env = dict(os.environ, SEMTOOLS_MODEL_DIR=model_dir, SEMTOOLS_TIMING="1")
    t0 = time.perf_counter()
    pr = subprocess.run([cli, "serve", *files, "-n", "1", "--top-k", "3", "--batch", "1024"], input="\n".join(many) + "\n", capture_output=True, text=True, env=env)
    wall = time.perf_counter() - t0
    ph = {}
    for line in pr.stderr.splitlines():
        if line.startswith('{"timing_ms"'):
            ph = json.loads(line)["timing_ms"]
    sess = sum(v for k, v in ph.items() if k.startswith("session_"))
    between = ph.get("between_session_calls", 0.0)
    result["cli_serve_batch_1024"] = {"queries": len(many), "returncode": pr.returncode, "wall_s": round(wall, 3), "stdout_bytes": len(pr.stdout),
                                      "phases_ms": {k: round(v, 1) for k, v in ph.items()},
                                      "queries_per_s_inside_session_calls": round(len(many) / (sess / 1e3)) if sess else None,
                                      "queries_per_s_incl_stdin_stdout": round(len(many) / ((sess + between) / 1e3)) if sess else None}
    print(json.dumps(result))
    if args.out:
        json.dump(result, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
