"""End-to-end wall time of the CLI replica (`semtools_amd/bin/semtools search`), split into phases (SEMTOOLS_TIMING=1):
  c1        1 query over ONE file of 1 000 lines, no workspace (BASELINE config c1's plumbing case)
  ws-cold   first search of a workspace of N lines (default 1 M, 100 files): tokenise || H2D || K1 + persist
  ws-warm   the same search again: table upload, corpus load through pinned buffers, scan, print
with a potion-sized synthetic model on disk (V = 500 000 rows x 256 f32 = 512 MB, like potion-multilingual-128M).
Also reports the host embedding pipeline alone (lines/s) with the native vocab tokenizer and with a Hugging Face
`tokenizers` WordPiece model plugged in as the callback tokenizer."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import synth  # noqa: E402

BIN = os.path.join(ROOT, "semtools_amd", "bin", "semtools")


def run_cli(args, env):
    t0 = time.perf_counter()
    p = subprocess.run([BIN] + args, env=env, capture_output=True, text=True)
    wall = time.perf_counter() - t0
    if p.returncode != 0:
        raise RuntimeError(p.stderr[-2000:])
    timing = {}
    for line in p.stderr.splitlines():
        if line.startswith('{"timing_ms"'):
            timing = json.loads(line)["timing_ms"]
    return wall, timing, p.stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vocab", type=int, default=500_000)
    ap.add_argument("--ws-lines", type=int, default=1_000_000)
    ap.add_argument("--ws-files", type=int, default=100)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from safetensors.numpy import save_file

    tmp = tempfile.mkdtemp(prefix="smt_cli_")
    model_dir = os.path.join(tmp, "model")
    os.makedirs(model_dir)
    V = args.vocab
    rng = np.random.default_rng(2)
    table = (rng.standard_normal((V, 256), dtype=np.float32) * np.float32(0.1))
    save_file({"embeddings": table}, os.path.join(model_dir, "model.safetensors"))
    with open(os.path.join(model_dir, "vocab.txt"), "w") as f:
        f.write("".join(f"w{i}\n" for i in range(V - 1)) + "[UNK]\n")
    json.dump({"normalize": True, "unk_token": "[UNK]"}, open(os.path.join(model_dir, "config.json"), "w"))
    pool = synth.pseudo_prose(20000, vocab_size=V - 1, seed=1)
    env = dict(os.environ, SEMTOOLS_MODEL_DIR=model_dir, HOME=tmp, SEMTOOLS_TIMING="1")
    env.pop("SEMTOOLS_WORKSPACE", None)
    result = dict(model=f"synthetic V={V} x 256 f32 ({V * 1024 / 1e6:.0f} MB) + vocab tokenizer", cases={})

    # ---- c1: one file of 1000 lines
    c1 = os.path.join(tmp, "c1.txt")
    open(c1, "w").write("\n".join(pool[:1000]) + "\n")
    runs = [run_cli(["search", pool[17], c1, "--top-k", "3", "-n", "3"], env) for _ in range(args.reps)]
    best = min(runs, key=lambda r: r[0])
    result["cases"]["c1_1k_lines"] = dict(wall_s=round(best[0], 4), phases_ms=best[1], first_line=best[2].split("\n")[0])
    print(json.dumps({"c1_1k_lines": result["cases"]["c1_1k_lines"]}), flush=True)

    # ---- workspace of ws_lines lines
    per = args.ws_lines // args.ws_files
    files = []
    for i in range(args.ws_files):
        p = os.path.join(tmp, f"doc{i:03d}.txt")
        with open(p, "w") as f:
            f.write("\n".join(pool[(i * 131 + j) % len(pool)] for j in range(per)) + "\n")
        files.append(p)
    run_cli(["workspace", "use", "bench"], env)
    wall, timing, out = run_cli(["search", pool[4242], *files, "--top-k", "3", "-n", "1"], dict(env, SEMTOOLS_WORKSPACE="bench"))
    result["cases"]["workspace_cold"] = dict(lines=per * args.ws_files, files=args.ws_files, wall_s=round(wall, 3), phases_ms=timing,
                                             lines_per_s=round(per * args.ws_files / (timing.get("embed_and_persist_changed_files", 1e9) / 1e3)))
    print(json.dumps({"workspace_cold": result["cases"]["workspace_cold"]}), flush=True)
    runs = [run_cli(["search", pool[4242], *files, "--top-k", "3", "-n", "1"], dict(env, SEMTOOLS_WORKSPACE="bench")) for _ in range(args.reps)]
    best = min(runs, key=lambda r: r[0])
    assert best[2] == out, "warm and cold answers differ"
    result["cases"]["workspace_warm"] = dict(lines=per * args.ws_files, wall_s=round(best[0], 3), phases_ms=best[1])
    print(json.dumps({"workspace_warm": result["cases"]["workspace_warm"]}), flush=True)

    # ---- the embedding pipeline alone: tokenise (host threads) || H2D || K1
    import torch  # noqa: F401
    import semtools_amd as smt
    from semtools_amd import host

    ctx = smt.Context(0)
    content = "\n".join(pool[i % len(pool)] for i in range(args.ws_lines)) + "\n"
    pipe = {}
    m = host.StaticModel(ctx, model_dir=model_dir)
    for rep in range(2):
        t0 = time.perf_counter()
        host.search_content(m, pool[17], content, n_lines=0, top_k=3)
        dt = time.perf_counter() - t0
    pipe["vocab_tokenizer_native"] = dict(lines=args.ws_lines, seconds=round(dt, 3), lines_per_s=round(args.ws_lines / dt))
    m.close()
    try:  # a real tokenizer.json (WordPiece over the same vocabulary) through the callback interface
        from tokenizers import Tokenizer, models, pre_tokenizers

        tok = Tokenizer(models.WordPiece({**{f"w{i}": i for i in range(V - 1)}, "[UNK]": V - 1}, unk_token="[UNK]"))
        tok.pre_tokenizer = pre_tokenizers.Whitespace()
        tok.save(os.path.join(model_dir, "tokenizer.json"))
        from semtools_amd import hf

        m3 = host.StaticModel(ctx, model_dir=model_dir)     # tokenizer.json present now: read NATIVELY by the C++ host
        for rep in range(2):
            t0 = time.perf_counter()
            host.search_content(m3, pool[17], content, n_lines=0, top_k=3)
            dt3 = time.perf_counter() - t0
        pipe["tokenizer_json_native_wordpiece"] = dict(lines=args.ws_lines, seconds=round(dt3, 3), lines_per_s=round(args.ws_lines / dt3))
        m3.close()
        # the CLI with that model directory (tokenizer.json read natively at every start): c1 and the warm workspace again
        runs = [run_cli(["search", pool[17], c1, "--top-k", "3", "-n", "3"], env) for _ in range(args.reps)]
        best = min(runs, key=lambda r: r[0])
        result["cases"]["c1_1k_lines_tokenizer_json"] = dict(wall_s=round(best[0], 4), phases_ms=best[1], first_line=best[2].split("\n")[0])
        runs = [run_cli(["search", pool[4242], *files, "--top-k", "3", "-n", "1"], dict(env, SEMTOOLS_WORKSPACE="bench")) for _ in range(args.reps)]
        best = min(runs, key=lambda r: r[0])
        result["cases"]["workspace_warm_tokenizer_json"] = dict(lines=per * args.ws_files, wall_s=round(best[0], 3), phases_ms=best[1])
        print(json.dumps({k: result["cases"][k] for k in ("c1_1k_lines_tokenizer_json", "workspace_warm_tokenizer_json")}), flush=True)
        m2 = hf.load_static_model(ctx, model_dir)
        n2 = min(args.ws_lines, 200_000)
        content2 = "\n".join(pool[i % len(pool)] for i in range(n2)) + "\n"
        t0 = time.perf_counter()
        host.search_content(m2, pool[17], content2, n_lines=0, top_k=3)
        dt = time.perf_counter() - t0
        pipe["hf_tokenizers_callback"] = dict(lines=n2, seconds=round(dt, 3), lines_per_s=round(n2 / dt),
                                              note="one Python call per line through the C callback: interpreter-bound")
        m2.close()
    except Exception as exc:  # noqa: BLE001
        pipe["hf_tokenizers_callback"] = dict(error=repr(exc))
    result["embed_pipeline"] = pipe
    print(json.dumps({"embed_pipeline": pipe}), flush=True)
    if args.out:
        json.dump(result, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
