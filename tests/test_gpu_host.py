"""GPU: the host layer end to end (config c1: `semtools search` of 1 query over 1k plaintext lines) --
create_document_from_content / search_documents / search_files / search_with_workspace and the
CLI replica -- compared BYTE FOR BYTE with the reference's output format filled with oracle numbers."""
import ctypes
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from tests import refimpl, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "semtools_amd", "bin", "semtools")
V = 20000


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    """A synthetic potion-style model on disk: model.safetensors (embeddings [V,256] f32), vocab.txt, config.json."""
    from safetensors.numpy import save_file

    d = tmp_path_factory.mktemp("model")
    table = synth.table(V, seed=2)
    save_file({"embeddings": table}, str(d / "model.safetensors"))
    (d / "vocab.txt").write_text("".join(f"w{i}\n" for i in range(V - 1)) + "[UNK]\n")
    (d / "config.json").write_text(json.dumps({"normalize": True, "unk_token": "[UNK]"}))
    return d, table


@pytest.fixture(scope="module", params=["one_gpu", "three_shards"])
def model(request, gpu_ctx, model_dir):
    """The host layer on one GPU, and on a group of three shards (VERDICT r2 row e': table replicated, lines and rows
    dealt over the shards, searches merged after one all-gather).  Every test that takes `model` runs both ways and
    compares with the SAME expected bytes, so sharded output == single-GPU output == the reference's format."""
    import semtools_amd as smt
    from semtools_amd import host

    group = None
    if request.param == "three_shards":
        group = smt.Group.logical(0, 3)       # (a 1-GPU box: device copies stand in for RCCL between the three ranks)
    m = host.StaticModel(group if group is not None else gpu_ctx, model_dir=model_dir[0])
    m.n_shards = 3 if group is not None else 1
    m.devices_env = {"SEMTOOLS_DEVICES": "0:3"} if group is not None else {}
    yield m
    m.close()
    if group is not None:
        group.close()


def tok(text):
    """the vocab tokenizer: whitespace words, unknown words -> [UNK] (dropped by the model, like model2vec)"""
    out = []
    for w in text.split():
        if w.startswith("w") and w[1:].isdigit() and int(w[1:]) < V - 1 and str(int(w[1:])) == w[1:]:
            out.append(int(w[1:]))
    return out


def oracle_embed(table, lines, max_tokens):
    ids, offsets = [], [0]
    for ln in lines:
        t = tok(ln)
        ids += t
        offsets.append(len(ids))
    return orc.embed_lines(table, np.array(ids, np.uint32), np.array(offsets, np.uint64), True, max_tokens)


def expected_results(table, docs, query, n_lines, top_k, max_distance=None, lower=False):
    emb = np.concatenate([oracle_embed(table, [l.lower() for l in lines] if lower else lines, 2048) for _, lines in docs])
    q = oracle_embed(table, [query.lower() if lower else query], 512)[0]
    res = orc.search_documents(emb, [len(l) for _, l in docs], q, n_lines, top_k, max_distance, accurate=True)
    return refimpl.results_from_oracle(docs, res)


@pytest.fixture(scope="module")
def prose_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("corpus")
    lines = synth.pseudo_prose(1000, vocab_size=V - 1, seed=1)
    lines[500] = ""                                   # an empty line embeds to the zero vector
    lines[600] = "unknownword anotherunknown"         # only unk tokens -> zero vector too
    f1 = d / "doc1.txt"
    f1.write_text("\n".join(lines) + "\n")
    lines2 = synth.pseudo_prose(37, vocab_size=V - 1, seed=7)
    f2 = d / "doc2.txt"
    f2.write_text("\r\n".join(lines2))               # CRLF, no trailing newline
    return [(str(f1), lines), (str(f2), lines2)]


def test_c1_search_one_query_over_1k_lines_text_and_json(model, model_dir, prose_files):
    from semtools_amd import host

    table = model_dir[1]
    docs = prose_files[:1]
    query = docs[0][1][17]
    want = expected_results(table, docs, query, 3, 3)
    assert want[0]["match_line"] == 17 and want[0]["distance"] < 1e-9
    got = host.search_files(model, query, [docs[0][0]], n_lines=3, top_k=3)
    assert got == refimpl.print_search_results(want)
    got_json = host.search_files(model, query, [docs[0][0]], n_lines=3, top_k=3, json=True)
    assert got_json == refimpl.search_results_json(want)
    assert host.search_files(model, query, [docs[0][0]], is_tty=True) == refimpl.print_search_results(want, is_tty=True)


@pytest.mark.parametrize("devices", [None, "0:3", "0"])
def test_cli_binary_matches_reference_format(model_dir, prose_files, devices):
    """`semtools search` on one GPU (default), on three shards ($SEMTOOLS_DEVICES: table replicated, lines dealt over the
    shards, one all-gather per search) and on an explicit one-GPU list: the same bytes."""
    table = model_dir[1]
    env = dict(os.environ, SEMTOOLS_MODEL_DIR=str(model_dir[0]))
    env.pop("SEMTOOLS_WORKSPACE", None)
    env.pop("SEMTOOLS_DEVICES", None)
    if devices:
        env["SEMTOOLS_DEVICES"] = devices
    files = [p for p, _ in prose_files]
    query = prose_files[1][1][5]
    want = expected_results(table, prose_files, query, 2, 4)
    r = subprocess.run([CLI, "search", query, *files, "-n", "2", "--top-k", "4"], capture_output=True, text=True, env=env,
                       stdin=subprocess.DEVNULL)
    assert r.returncode == 0, r.stderr
    assert r.stdout == refimpl.print_search_results(want)
    r = subprocess.run([CLI, "search", query, *files, "--context", "2", "--top-k", "4", "--json"], capture_output=True,
                       text=True, env=env, stdin=subprocess.DEVNULL)
    assert r.stdout == refimpl.search_results_json(want)
    # threshold mode: every hit under the distance, top_k ignored (src/search/mod.rs:115-116)
    want_thr = expected_results(table, prose_files, query, 0, 3, max_distance=0.8)
    r = subprocess.run([CLI, "search", query, *files, "-n", "0", "-m", "0.8"], capture_output=True, text=True, env=env,
                       stdin=subprocess.DEVNULL)
    assert r.stdout == refimpl.print_search_results(want_thr) and len(want_thr) > 3
    # stdin branch (src/cmds/search.rs:145-176)
    content = "\n".join(prose_files[1][1]) + "\n"
    want_stdin = expected_results(table, [("<stdin>", prose_files[1][1])], query, 1, 2)
    r = subprocess.run([CLI, "search", query, "-n", "1", "--top-k", "2"], input=content, capture_output=True, text=True, env=env)
    assert r.stdout == refimpl.print_search_results(want_stdin)
    # no input at all
    r = subprocess.run([CLI, "search", query], capture_output=True, text=True, env=env, stdin=subprocess.DEVNULL)
    assert r.returncode == 1 and "No input provided" in r.stderr and r.stdout == ""
    # unreadable file aborts the whole command (src/search/mod.rs:130)
    r = subprocess.run([CLI, "search", query, files[0], "/nonexistent/file.txt"], capture_output=True, text=True, env=env,
                       stdin=subprocess.DEVNULL)
    assert r.returncode == 1 and r.stdout == ""
    # a malformed device list is an error, not a silent single GPU
    r = subprocess.run([CLI, "search", query, files[0]], capture_output=True, text=True, env=dict(env, SEMTOOLS_DEVICES="0;1"),
                       stdin=subprocess.DEVNULL)
    assert r.returncode == 1 and "device spec" in r.stderr


def test_cli_workspace_is_interchangeable_between_one_gpu_and_three_shards(model_dir, prose_files, tmp_path):
    """A workspace filled by a 3-shard run is searched by a 1-GPU run and extended by it, then searched by 3 shards again
    (the vector file holds global row order; line_rows.json records how 3 shards dealt the rows and a group of another size
    re-cuts them): every run prints the same bytes."""
    env = dict(os.environ, SEMTOOLS_MODEL_DIR=str(model_dir[0]), HOME=str(tmp_path))
    env.pop("SEMTOOLS_WORKSPACE", None)
    env.pop("SEMTOOLS_DEVICES", None)
    three = dict(env, SEMTOOLS_DEVICES="0:3")
    files = [p for p, _ in prose_files]
    query = prose_files[0][1][250]

    def run(e, fs, *extra):
        r = subprocess.run([CLI, "search", query, *fs, "-w", "mix", "-n", "1", "--top-k", "5", *extra], capture_output=True, text=True,
                           env=e, stdin=subprocess.DEVNULL)
        assert r.returncode == 0, r.stderr
        return r.stdout, r.stderr

    assert subprocess.run([CLI, "workspace", "use", "mix"], env=env, capture_output=True).returncode == 0
    first, err = run(three, files[:1])
    assert "Updating workspace with 1000 lines" in err
    rows = json.loads((tmp_path / ".semtools/workspaces/mix/line_rows.json").read_text())
    assert rows["shards"]["n_ranks"] == 3 and sum(p[0] for p in rows["shards"]["pieces"]) == 1000
    again, err = run(env, files[:1])                                    # one GPU reads what three wrote
    assert again == first and "Updating" not in err
    both, err = run(env, files)                                         # ... and appends a document
    assert "Updating workspace with 37 lines" in err
    both3, err = run(three, files)                                      # three shards read what one wrote
    assert both3 == both and "Updating" not in err
    js1, _ = run(env, files, "--json")
    js3, _ = run(three, files, "--json")
    assert js1 == js3 and json.loads(js1)["results"][0]["match_line_number"] == 250
    st = subprocess.run([CLI, "workspace", "status", "mix"], env=three, capture_output=True, text=True)
    assert st.returncode == 0 and "Documents: 2" in st.stdout


def test_context_window_clamps_at_file_boundaries(model, model_dir, tmp_path):
    from semtools_amd import host

    f = tmp_path / "small.txt"
    f.write_text("w1 w2\nw3 w4")
    docs = [(str(f), ["w1 w2", "w3 w4"])]
    want = expected_results(model_dir[1], docs, "w1 w2", 5, 3)
    assert want[0]["start"] == 0 and want[0]["end"] == 2                      # mod.rs:337-357
    assert host.search_files(model, "w1 w2", [str(f)], n_lines=5, top_k=3) == refimpl.print_search_results(want)
    e = tmp_path / "empty.txt"
    e.write_text("")
    assert host.search_files(model, "w1", [str(e)]) == ""                      # empty content -> no Document


def test_ignore_case_embeds_lowercase_but_prints_original(model, model_dir, tmp_path):
    from semtools_amd import host

    f = tmp_path / "mixed.txt"
    lines = ["W10 W11 w12", "w20 W21", "W30"]
    f.write_text("\n".join(lines))
    want = expected_results(model_dir[1], [(str(f), lines)], "W20 w21", 0, 3, lower=True)
    got = host.search_files(model, "W20 w21", [str(f)], n_lines=0, top_k=3, ignore_case=True)
    assert got == refimpl.print_search_results(want)
    assert "   2: w20 W21" in got and want[0]["match_line"] == 1 and want[0]["distance"] < 1e-9
    # without -i the upper-case words are unknown tokens
    plain = host.search_files(model, "W20 w21", [str(f)], n_lines=0, top_k=1)
    assert plain == refimpl.print_search_results(expected_results(model_dir[1], [(str(f), lines)], "W20 w21", 0, 1))


def test_encode_matches_oracle_bit_for_bit(model, model_dir):
    sents = ["w1 w2 w3", "", "zzz", "w5 " * 600, "w7\tw8  w9"]
    got = model.encode_with_args(sents, 2048)
    assert np.array_equal(got, oracle_embed(model_dir[1], sents, 2048))
    assert np.array_equal(model.encode_single("w5 " * 600), oracle_embed(model_dir[1], ["w5 " * 600], 512)[0])


def test_callback_tokenizer_plugs_in(gpu_ctx, model_dir):
    from semtools_amd import host

    m = host.StaticModel(gpu_ctx, table=model_dir[1], tokenizer=lambda t: tok(t) + [V - 1], unk_id=V - 1)
    assert np.array_equal(m.encode_with_args(["w3 w4 nope"], 2048), oracle_embed(model_dir[1], ["w3 w4"], 2048))
    m.close()


def test_lazy_table_gives_the_same_embeddings_as_the_full_upload(gpu_ctx, model_dir, monkeypatch):
    """A file-backed model uploads nothing until an embed call shows what it needs: small calls pool from a compact
    table of just the rows they touch, a large call (> 32768 lines, or ids covering > 1/16 of the table) uploads the
    whole table once.  All three routes -- compact, full after lazy, eager -- give bit-identical rows."""
    from semtools_amd import host

    lines = synth.pseudo_prose(40000, vocab_size=V - 1, seed=5)
    lazy = host.StaticModel(gpu_ctx, model_dir=model_dir[0])
    small = lazy.encode_with_args(lines[:300], 2048)                       # compact table (300 lines touch < V/16 ids)
    assert np.array_equal(small, oracle_embed(model_dir[1], lines[:300], 2048))
    q = lazy.encode_with_args([lines[7]], 512)                              # one line: a handful of rows
    assert np.array_equal(q[0], small[7])
    big = lazy.encode_with_args(lines, 2048)                                # covers most of the vocabulary: full upload
    after = lazy.encode_with_args(lines[:300], 2048)                        # ... which later small calls then use
    lazy.close()
    monkeypatch.setenv("SEMTOOLS_EAGER_MODEL", "1")
    eager = host.StaticModel(gpu_ctx, model_dir=model_dir[0])
    monkeypatch.delenv("SEMTOOLS_EAGER_MODEL")
    ref = eager.encode_with_args(lines, 2048)
    eager.close()
    assert np.array_equal(big, ref) and np.array_equal(after, ref[:300]) and np.array_equal(small, ref[:300])


def test_many_small_files_plain_and_workspace(gpu_ctx, model_dir, tmp_path, monkeypatch, capfd):
    """A repository is hundreds of small files.  `search_files` and the workspace embed all of them through ONE pipeline
    run (the workspace logs one token record per file); a long series of separate small embed calls makes the lazy
    model switch from compact tables to the full table (after 64).  Same answers as the oracle on every route."""
    from semtools_amd import host

    monkeypatch.setenv("HOME", str(tmp_path))
    monkeypatch.delenv("SEMTOOLS_WORKSPACE", raising=False)
    pool = synth.pseudo_prose(400, vocab_size=V - 1, seed=19)
    docs = []
    for i in range(90):
        lines = pool[4 * i: 4 * i + 1 + i % 4]
        f = tmp_path / f"src_{i:03d}.txt"
        f.write_text("\n".join(lines) + "\n")
        docs.append((str(f), lines))
    files = [p for p, _ in docs]
    query = docs[71][1][0]
    fresh = host.StaticModel(gpu_ctx, model_dir=model_dir[0])
    try:
        for i in range(70):                                             # lazy: 70 small calls cross the 64-call switch
            assert np.array_equal(fresh.encode_with_args(docs[i][1], 2048), oracle_embed(model_dir[1], docs[i][1], 2048)), i
        want = expected_results(model_dir[1], docs, query, 0, 5)
        assert host.search_files(fresh, query, files, n_lines=0, top_k=5) == refimpl.print_search_results(want)
        host.workspace_use(None, "many")
        out = host.search_with_workspace(fresh, query, files, workspace_name="many", n_lines=0, top_k=5)
        assert f"Updating workspace with {sum(len(l) for _, l in docs)} lines" in capfd.readouterr().err
        assert out.split("\n")[0].startswith(f"{files[71]}:0::1 (")
        root = tmp_path / ".semtools" / "workspaces" / "many"
        log = (root / "line_tokens.log").read_bytes()
        assert log.count(b"TOKD") == 90                                # one record per file
        txt = host.workspace_reembed(fresh, "many")
        assert txt.startswith(f"Re-embedded {sum(len(l) for _, l in docs)} lines of 90 documents")
        assert host.search_with_workspace(fresh, query, files, workspace_name="many", n_lines=0, top_k=5) == out
    finally:
        fresh.close()


def test_workspace_flow(model, model_dir, prose_files, tmp_path, monkeypatch, capfd):
    """search_with_workspace (src/search/mod.rs:146-216): first run embeds and persists, second run reuses,
    a modified file is re-embedded, prune drops deleted files, status/stats keep the reference's text."""
    from semtools_amd import host

    monkeypatch.setenv("HOME", str(tmp_path))
    monkeypatch.delenv("SEMTOOLS_WORKSPACE", raising=False)
    table = model_dir[1]
    a = tmp_path / "a.txt"
    b = tmp_path / "b.txt"
    la = prose_files[1][1][:20]
    lb = prose_files[1][1][20:37]
    a.write_text("\n".join(la) + "\n")
    b.write_text("\n".join(lb) + "\n")
    files = [str(a), str(b)]
    host.workspace_use(None, "t1")
    query = lb[3]

    def want(docs, k, thr=None):
        emb = np.concatenate([oracle_embed(table, lines, 2048) for _, lines in docs])
        path = np.concatenate([np.full(len(l), i, np.uint32) for i, (_, l) in enumerate(docs)])
        line = np.concatenate([np.arange(len(l), dtype=np.int32) for _, l in docs])
        q = oracle_embed(table, [query], 512)[0]
        return orc.search_line_embeddings(emb, path, line, q, np.arange(len(docs), dtype=np.uint32), k, thr)

    def parse(text):
        hits = []
        for blk in text.strip("\n").split("\n\n"):
            head = blk.split("\n")[0]
            name, rest = head.rsplit(":", 3)[0], head[len(head.rsplit(":", 3)[0]) + 1:]
            start, rest = rest.split("::")
            end, dist = rest.split(" (")
            hits.append((name, int(start), int(end), float(dist[:-1]), blk.split("\n")[1:]))
        return hits

    out = host.search_with_workspace(model, query, files, workspace_name="t1", n_lines=1, top_k=3)
    err = capfd.readouterr().err
    assert "Updating workspace with 37 lines from new/changed docs..." in err
    assert "Updating workspace with 2 new/changed documents..." in err
    exp = want([(str(a), la), (str(b), lb)], 3)
    hits = parse(out)
    assert [(h[0], h[1], h[2]) for h in hits] == [(files[r["path_id"]], max(r["line_number"] - 1, 0), r["line_number"] + 2) for r in exp]
    assert hits[0][0] == str(b) and hits[0][1] == 2 and hits[0][2] == 5            # header end is NOT clamped
    np.testing.assert_allclose([h[3] for h in hits], [r["distance"] for r in exp], atol=2e-6)
    assert hits[0][4] == [f"{i + 1:4}: {lb[i]}" for i in (2, 3, 4)]

    out2 = host.search_with_workspace(model, query, files, workspace_name="t1", n_lines=1, top_k=3)
    assert capfd.readouterr().err == "" and out2 == out                              # unchanged -> nothing re-embedded
    # (the reference prints a hard-coded "Index: Yes (HNSW)" although it scans exactly; here a small workspace says
    # what it does -- the exact scan -- and a large one names its IVF index, see test_workspace_index_lifecycle)
    assert "Documents: 2" in host.workspace_status(model.ctx, "t1") and "Index: No" in host.workspace_status(model.ctx, "t1")
    assert json.loads(host.workspace_status(model.ctx, "t1", json=True))["total_documents"] == 2

    # workspace mode: top_k applies even with a threshold (store.rs:543)
    out3 = host.search_with_workspace(model, query, files, workspace_name="t1", n_lines=0, top_k=2, max_distance=0.99)
    assert len(parse(out3)) == len(want([(str(a), la), (str(b), lb)], 2, 0.99)) <= 2

    # modify b (shrinks): re-embedded, no stale tail rows survive
    lb2 = lb[:5]
    b.write_text("\n".join(lb2) + "\n")
    os.utime(b, (1_900_000_000, 1_900_000_000))
    out4 = host.search_with_workspace(model, query, files, workspace_name="t1", n_lines=0, top_k=30)
    assert "Updating workspace with 5 lines" in capfd.readouterr().err
    assert len(parse(out4)) == 25 and all(h[1] < 5 for h in parse(out4) if h[0] == str(b))

    # JSON output of workspace mode
    js = json.loads(host.search_with_workspace(model, query, files, workspace_name="t1", n_lines=1, top_k=1, json=True))
    r0 = js["results"][0]
    assert list(r0) == ["filename", "start_line_number", "end_line_number", "match_line_number", "distance", "content"]
    assert r0["filename"] == str(b) and r0["match_line_number"] == 3 and r0["content"] == "\n".join(lb2[2:5])

    # delete a, prune
    a.unlink()
    txt = host.workspace_prune(model.ctx, "t1")
    assert txt == f"Found 1 stale documents:\n  - {a}\nRemoved 1 stale documents from workspace.\n"
    assert host.workspace_prune(model.ctx, "t1") == "No stale documents found. Workspace is clean.\n"
    assert json.loads(host.workspace_prune(model.ctx, "t1", json=True)) == {"files_removed": 0, "files_remaining": 1}
    out5 = host.search_with_workspace(model, query, files, workspace_name="t1", n_lines=0, top_k=30)
    assert all(h[0] == str(b) for h in parse(out5)) and len(parse(out5)) == 5


def test_workspace_reembed_from_cached_tokens(gpu_ctx, model, model_dir, prose_files, tmp_path, monkeypatch, capfd):
    """the workspace keeps the token ids it pooled (line_tokens.log); a new embedding table behind the
    same tokenizer re-embeds the store on the GPU without reading a source file.  Vectors = the oracle's for the new
    table, bit for bit; a different tokenizer is refused; a store without cached tokens is left alone."""
    import semtools_amd as smt
    from semtools_amd import host

    monkeypatch.setenv("HOME", str(tmp_path))
    monkeypatch.delenv("SEMTOOLS_WORKSPACE", raising=False)
    a, b = tmp_path / "a.txt", tmp_path / "b.txt"
    la, lb = prose_files[1][1][:15], prose_files[0][1][495:505] + prose_files[0][1][598:603]   # incl. an empty and an all-unk line
    a.write_text("\n".join(la) + "\n")
    b.write_text("\n".join(lb) + "\n")
    files = [str(a), str(b)]
    host.workspace_use(None, "tk")
    host.search_with_workspace(model, la[2], files, workspace_name="tk", n_lines=0, top_k=3)
    # replace a (its first record in the log is superseded)
    la = la[:9] + ["w5 w6 w7 w5"]
    a.write_text("\n".join(la) + "\n")
    os.utime(a, (1_900_000_000, 1_900_000_000))
    host.search_with_workspace(model, la[2], files, workspace_name="tk", n_lines=0, top_k=3)
    capfd.readouterr()
    root = tmp_path / ".semtools" / "workspaces" / "tk"
    assert (root / "line_tokens.log").exists()

    table2 = synth.table(V, seed=77)                                   # "the next version of the model": same tokenizer
    model2 = host.StaticModel(gpu_ctx, table=table2, tokenizer=("vocab", model_dir[0] / "vocab.txt", "[UNK]"))
    try:
        a.rename(tmp_path / "a.away")                                   # no source file is needed ...
        b.rename(tmp_path / "b.away")
        txt = host.workspace_reembed(model2, "tk")
        assert txt == f"Re-embedded {len(la) + len(lb)} lines of 2 documents from cached tokens ({sum(len(tok(l)) for l in la + lb)} tokens).\n"
        (tmp_path / "a.away").rename(a)
        (tmp_path / "b.away").rename(b)
        os.utime(a, (1_900_000_000, 1_900_000_000))
        c = smt.Corpus.load(gpu_ctx, str(root / "line_embeddings.f32"))
        got = c.read_rows(0, c.rows)
        c.close()
        rows = json.loads((root / "line_rows.json").read_text())["extents"]
        ext = {e["path"]: (e["first_row"], e["n_rows"]) for e in rows}
        assert ext[str(a)][1] == len(la) and ext[str(b)][1] == len(lb) and len(got) == len(la) + len(lb)   # compacted: live rows only
        for path, lines in ((str(a), la), (str(b), lb)):
            f0, n = ext[path]
            assert np.array_equal(got[f0:f0 + n], oracle_embed(table2, lines, 2048)), path
        # ... and the next search with the new model finds everything unchanged (nothing re-embedded) and uses the new vectors
        out = host.search_with_workspace(model2, lb[3], files, workspace_name="tk", n_lines=0, top_k=2)
        assert capfd.readouterr().err == ""
        assert out.split("\n")[0].startswith(f"{b}:3::4 (")
        js = json.loads(host.workspace_reembed(model2, "tk", json=True))
        assert js["documents_reembedded"] == 2 and js["documents_without_cached_tokens"] == []

        # the CLI form (model from $SEMTOOLS_MODEL_DIR: back to the first table)
        env = dict(os.environ, SEMTOOLS_MODEL_DIR=str(model_dir[0]))
        r = subprocess.run([CLI, "workspace", "reembed", "tk"], env=env, capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith(f"Re-embedded {len(la) + len(lb)} lines of 2 documents"), r.stderr
        ext = {e["path"]: e["first_row"] for e in json.loads((root / "line_rows.json").read_text())["extents"]}
        c = smt.Corpus.load(gpu_ctx, str(root / "line_embeddings.f32"))
        assert np.array_equal(c.read_rows(ext[str(a)], len(la)), oracle_embed(model_dir[1], la, 2048))
        c.close()

        # a model with another tokenizer is refused
        other = host.StaticModel(gpu_ctx, table=table2, tokenizer="hash")
        with pytest.raises(Exception, match="different tokenizer"):
            host.workspace_reembed(other, "tk")
        other.close()

        # a workspace filled with the cache turned off: reported, untouched
        monkeypatch.setenv("SEMTOOLS_TOKEN_CACHE", "0")
        host.workspace_use(None, "nocache")
        host.search_with_workspace(model, la[2], files, workspace_name="nocache", n_lines=0, top_k=3)
        monkeypatch.delenv("SEMTOOLS_TOKEN_CACHE")
        root2 = tmp_path / ".semtools" / "workspaces" / "nocache"
        before = (root2 / "line_embeddings.f32").read_bytes()
        txt = host.workspace_reembed(model2, "nocache")
        assert txt.startswith("No cached tokens for 2 documents (nothing was changed):") and f"  - {a}\n" in txt
        assert (root2 / "line_embeddings.f32").read_bytes() == before
    finally:
        model2.close()


def test_workspace_survives_a_damaged_store(model, model_dir, prose_files, tmp_path, monkeypatch, capfd):
    """Crash consistency: a truncated line_embeddings.f32, or metadata whose line rows are gone, must
    lead to a re-embed -- not to a dead workspace, and not to documents that silently drop out of the search."""
    from semtools_amd import host

    monkeypatch.setenv("HOME", str(tmp_path))
    monkeypatch.delenv("SEMTOOLS_WORKSPACE", raising=False)
    a, b = tmp_path / "a.txt", tmp_path / "b.txt"
    la, lb = prose_files[1][1][:12], prose_files[1][1][12:30]
    a.write_text("\n".join(la) + "\n")
    b.write_text("\n".join(lb) + "\n")
    files = [str(a), str(b)]
    host.workspace_use(None, "dmg")
    good = host.search_with_workspace(model, lb[4], files, workspace_name="dmg", n_lines=0, top_k=5)
    capfd.readouterr()
    root = tmp_path / ".semtools" / "workspaces" / "dmg"
    emb = root / "line_embeddings.f32"
    assert emb.exists() and not (root / "line_embeddings.f32.tmp").exists()
    # (1) the vectors file loses its tail (crash / ENOSPC during a rewrite)
    data = emb.read_bytes()
    emb.write_bytes(data[: len(data) // 2])
    again = host.search_with_workspace(model, lb[4], files, workspace_name="dmg", n_lines=0, top_k=5)
    err = capfd.readouterr().err
    assert "is unreadable" in err and "Updating workspace with 30 lines from new/changed docs..." in err
    assert again == good
    # (2) the extent table forgets one document while its metadata still says "unchanged"
    rows = json.loads((root / "line_rows.json").read_text())
    rows["extents"] = [e for e in rows["extents"] if e["path"] != str(b)]
    (root / "line_rows.json").write_text(json.dumps(rows))
    third = host.search_with_workspace(model, lb[4], files, workspace_name="dmg", n_lines=0, top_k=5)
    assert "Updating workspace with 18 lines from new/changed docs..." in capfd.readouterr().err
    assert third == good
    # (3) repeated edits do not grow the matrix without bound: dead rows are compacted before the flush
    for i in range(6):
        b.write_text("\n".join(lb) + f"\nedit {i}\n")
        os.utime(b, (1_800_000_000 + i, 1_800_000_000 + i))
        host.search_with_workspace(model, lb[4], files, workspace_name="dmg", n_lines=0, top_k=5)
    capfd.readouterr()
    assert json.loads(host.workspace_status(model.ctx, "dmg", json=True))["total_documents"] == 2
    live = len(la) + len(lb) + 1
    assert (emb.stat().st_size - 32) // 1024 <= 4096 + 2 * live   # bounded (compaction threshold: 4096 dead rows or half)


def test_workspace_index_lifecycle(model, tmp_path, monkeypatch, capfd):
    """The IVF index of a large workspace: built by the first whole-workspace search above the
    row threshold, persisted beside the vectors, reloaded by the next process, extended incrementally when a file is
    added, bypassed for path subsets -- and `workspace status` names it.  With every list probed and more re-scored
    rows than a list holds the approximate path returns exactly what the exact scan returns."""
    from semtools_amd import host

    monkeypatch.setenv("HOME", str(tmp_path))
    monkeypatch.delenv("SEMTOOLS_WORKSPACE", raising=False)
    files = []
    for i in range(6):
        f = tmp_path / f"big{i}.txt"
        f.write_text("\n".join(synth.pseudo_prose(1000, vocab_size=V - 1, seed=100 + i)) + "\n")
        files.append(str(f))
    query = synth.pseudo_prose(1, vocab_size=V - 1, seed=103)[0]
    host.workspace_use(None, "big")
    root = tmp_path / ".semtools" / "workspaces" / "big"
    monkeypatch.setenv("SEMTOOLS_INDEX_MIN_ROWS", "1000000000")                 # exact scan: the truth
    exact = host.search_with_workspace(model, query, files, workspace_name="big", n_lines=0, top_k=5)
    n = model.n_shards
    parts = [root / ("line_index.ivf" if n == 1 else f"line_index.ivf.r{r}of{n}") for r in range(n)]
    assert not any(p.exists() for p in parts)
    assert "Index: No" in host.workspace_status(model.ctx, "big")
    monkeypatch.setenv("SEMTOOLS_INDEX_MIN_ROWS", "4000")
    monkeypatch.setenv("SEMTOOLS_INDEX_NPROBE", "512")
    capfd.readouterr()
    got = host.search_with_workspace(model, query, files, workspace_name="big", n_lines=0, top_k=5)
    assert got == exact
    assert all(p.exists() for p in parts) and not list(root.glob("line_index.ivf*.tmp"))
    assert "Index: Yes (IVF_PQ)" in host.workspace_status(model.ctx, "big")
    stamp = [p.stat().st_mtime_ns for p in parts]
    assert host.search_with_workspace(model, query, files, workspace_name="big", n_lines=0, top_k=5) == exact
    assert [p.stat().st_mtime_ns for p in parts] == stamp                        # reloaded, not rebuilt
    # a path subset is answered by the exact range-filtered scan
    sub = host.search_with_workspace(model, query, files[:2], workspace_name="big", n_lines=0, top_k=5)
    monkeypatch.setenv("SEMTOOLS_INDEX_MIN_ROWS", "1000000000")
    assert sub == host.search_with_workspace(model, query, files[:2], workspace_name="big", n_lines=0, top_k=5)
    # a new file: its rows are inserted into the existing lists (the index file is rewritten, not retrained)
    f = tmp_path / "big6.txt"
    f.write_text("\n".join(synth.pseudo_prose(800, vocab_size=V - 1, seed=106)) + "\n")
    files.append(str(f))
    exact7 = host.search_with_workspace(model, query, files, workspace_name="big", n_lines=0, top_k=5)
    monkeypatch.setenv("SEMTOOLS_INDEX_MIN_ROWS", "4000")
    size_before = sum(p.stat().st_size for p in parts)
    got7 = host.search_with_workspace(model, query, files, workspace_name="big", n_lines=0, top_k=5)
    assert got7 == exact7
    assert sum(p.stat().st_size for p in parts) == size_before + 800 * 36       # 32 B code + 4 B row id per new row
    # an edited file leaves dead rows behind: they never surface
    (tmp_path / "big0.txt").write_text("\n".join(synth.pseudo_prose(900, vocab_size=V - 1, seed=100)) + "\n")
    os.utime(tmp_path / "big0.txt", (1_950_000_000, 1_950_000_000))
    got8 = host.search_with_workspace(model, query, files, workspace_name="big", n_lines=0, top_k=5)
    monkeypatch.setenv("SEMTOOLS_INDEX_MIN_ROWS", "1000000000")
    assert got8 == host.search_with_workspace(model, query, files, workspace_name="big", n_lines=0, top_k=5)


def test_index_files_of_another_gpu_count_are_never_loaded(gpu_ctx, model_dir, tmp_path, monkeypatch, capfd):
    """index files are named per rank count (`line_index.ivf` vs `line_index.ivf.r<r>of<n>`).  A session with another
    number of GPUs that moves or rewrites rows must take EVERY index file with it, and an index is only ever loaded when
    `line_index.gen` names this corpus generation and this rank count -- a stale one would pass the loader's range checks and
    lose recall silently."""
    import semtools_amd as smt
    from semtools_amd import host

    monkeypatch.setenv("HOME", str(tmp_path))
    monkeypatch.delenv("SEMTOOLS_WORKSPACE", raising=False)
    files = []
    for i in range(6):
        f = tmp_path / f"gen{i}.txt"
        f.write_text("\n".join(synth.pseudo_prose(1000, vocab_size=V - 1, seed=300 + i)) + "\n")
        files.append(str(f))
    query = synth.pseudo_prose(1, vocab_size=V - 1, seed=303)[0]
    host.workspace_use(None, "gen")
    root = tmp_path / ".semtools" / "workspaces" / "gen"
    one = host.StaticModel(gpu_ctx, model_dir=model_dir[0])
    group = smt.Group.logical(0, 3)
    three = host.StaticModel(group, model_dir=model_dir[0])
    try:
        monkeypatch.setenv("SEMTOOLS_INDEX_MIN_ROWS", "1000000000")
        exact = host.search_with_workspace(one, query, files, workspace_name="gen", n_lines=0, top_k=5)
        monkeypatch.setenv("SEMTOOLS_INDEX_MIN_ROWS", "4000")
        monkeypatch.setenv("SEMTOOLS_INDEX_NPROBE", "512")
        assert host.search_with_workspace(one, query, files, workspace_name="gen", n_lines=0, top_k=5) == exact
        gen1 = json.loads((root / "line_index.gen").read_text())
        assert (root / "line_index.ivf").exists() and gen1["n_ranks"] == 1
        assert json.loads((root / "line_rows.json").read_text())["generation"] == gen1["generation"]
        # three shards on the same workspace: the one-GPU index is not theirs -- they build and name their own
        assert host.search_with_workspace(three, query, files, workspace_name="gen", n_lines=0, top_k=5) == exact
        gen3 = json.loads((root / "line_index.gen").read_text())
        assert gen3["n_ranks"] == 3 and all((root / f"line_index.ivf.r{r}of3").exists() for r in range(3))
        # ... and the one-GPU session no longer trusts the file it left behind (the sidecar names three ranks): rebuilt, same answer
        stale = root / "line_index.ivf"
        before = stale.stat().st_mtime_ns if stale.exists() else None
        assert host.search_with_workspace(one, query, files, workspace_name="gen", n_lines=0, top_k=5) == exact
        assert json.loads((root / "line_index.gen").read_text())["n_ranks"] == 1
        assert before is None or stale.stat().st_mtime_ns != before
        # rows rewritten by the three-shard session (re-embed from the token cache): a new generation, EVERY index file gone
        capfd.readouterr()
        host.workspace_reembed(three, "gen")
        assert not list(root.glob("line_index.*")), sorted(p.name for p in root.iterdir())
        assert json.loads((root / "line_rows.json").read_text())["generation"] > gen1["generation"]
        assert host.search_with_workspace(one, query, files, workspace_name="gen", n_lines=0, top_k=5) == exact
    finally:
        three.close()
        group.close()
        one.close()


def test_resident_session_and_serve_mode(model, model_dir, prose_files):
    """Batched-query surface: one embedding pass, many queries; every answer equals what a
    one-shot `semtools search` prints -- for small batches (K2) and batches >= 8 queries (K3 MFMA path)."""
    from semtools_amd import host

    files = [p for p, _ in prose_files]
    queries = [prose_files[0][1][i] for i in (3, 17, 99, 250, 400, 512, 640, 777, 801, 930, 998)] + [prose_files[1][1][5]]
    s = host.Session(model, files)
    assert s.lines == 1037
    for qs in (queries[:3], queries):                           # 3 queries -> K2, 12 queries -> K3
        got = s.search(qs, n_lines=2, top_k=4)
        for q, text in zip(qs, got):
            assert text == host.search_files(model, q, files, n_lines=2, top_k=4)
    thr = s.search(queries[:2], n_lines=0, max_distance=0.8)
    assert thr[0] == host.search_files(model, queries[0], files, n_lines=0, max_distance=0.8)
    js = s.search(queries[:9], n_lines=1, top_k=2, json=True)
    assert js[4] == host.search_files(model, queries[4], files, n_lines=1, top_k=2, json=True)
    s.close()
    env = dict(os.environ, SEMTOOLS_MODEL_DIR=str(model_dir[0]), **model.devices_env)
    r = subprocess.run([CLI, "serve", *files, "-n", "2", "--top-k", "4", "--batch", "16"], input="\n".join(queries) + "\n",
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    want = "".join(f"### {q}\n" + host.search_files(model, q, files, n_lines=2, top_k=4) for q in queries)
    assert r.stdout == want and "1037 lines resident" in r.stderr


def test_real_hf_tokenizer_json_plugs_into_the_host_layer(gpu_ctx, tmp_path):
    """A model2vec-style directory with a REAL tokenizer.json (WordPiece built with the HF `tokenizers` package,
    BERT normaliser/pre-tokeniser, [UNK]) + model.safetensors: the host layer must embed exactly what the
    reference pipeline embeds -- HF ids (no special tokens) -> drop unk -> truncate -> pool."""
    from safetensors.numpy import save_file
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, trainers
    from semtools_amd import hf, host

    corpus_lines = synth.pseudo_prose(400, vocab_size=300, seed=3) + ["Hello World, semantic search!", "naïve café déjà vu"]
    tok = Tokenizer(models.WordPiece(unk_token="[UNK]"))
    tok.normalizer = normalizers.BertNormalizer(lowercase=True)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.train_from_iterator(corpus_lines, trainers.WordPieceTrainer(vocab_size=400, special_tokens=["[PAD]", "[UNK]"]))
    d = tmp_path / "hfmodel"
    d.mkdir()
    tok.save(str(d / "tokenizer.json"))
    V = tok.get_vocab_size()
    table = synth.table(V, seed=5)
    save_file({"embeddings": table}, str(d / "model.safetensors"))
    (d / "config.json").write_text(json.dumps({"normalize": True}))
    m = hf.load_static_model(gpu_ctx, str(d))
    unk = tok.token_to_id("[UNK]")
    sents = ["Hello World, semantic search!", "w1 w2 zzzzqqq w3", "", "naïve café déjà vu", "🙂 emoji only"]
    ids, offsets = [], [0]
    for s in sents:
        t = [i for i in tok.encode(s, add_special_tokens=False).ids if i != unk][:2048]
        ids += t
        offsets.append(len(ids))
    want = orc.embed_lines(table, np.array(ids, np.uint32), np.array(offsets, np.uint64), True, 2048)
    assert np.array_equal(m.encode_with_args(sents, 2048), want)
    f = tmp_path / "doc.txt"
    f.write_text("\n".join(corpus_lines))
    out = host.search_files(m, corpus_lines[7], [str(f)], n_lines=0, top_k=1)
    assert out.startswith(f"{f}:7::8 (") and float(out.split("(")[1].split(")")[0]) < 1e-9
    m.close()
    # the same directory WITHOUT Python in the loop: smt_host_model_from_dir reads tokenizer.json natively
    # (hf_tokenizer.cpp; pinned against the `tokenizers` wheel by tests/test_tokenizer.py) and streams the table
    m2 = host.StaticModel(gpu_ctx, model_dir=str(d))
    assert np.array_equal(m2.encode_with_args(sents, 2048), want)
    assert host.search_files(m2, corpus_lines[7], [str(f)], n_lines=0, top_k=1) == out
    m2.close()
    env = dict(os.environ, SEMTOOLS_MODEL_DIR=str(d))
    cli = subprocess.run([os.path.join(ROOT, "semtools_amd", "bin", "semtools"), "search", corpus_lines[7], str(f), "-n", "0", "--top-k", "1"],
                         env=env, capture_output=True, text=True)
    assert cli.returncode == 0 and cli.stdout == out, cli.stderr


@pytest.mark.parametrize("dtype", ["float16", "int8"])
def test_model_tables_stored_as_f16_or_i8_are_widened_like_model2vec(gpu_ctx, tmp_path, dtype):
    """StaticModel::from_pretrained accepts F32 / F16 / I8 `embeddings` and widens to f32: the
    embeddings must equal the oracle's on the widened table, bit for bit."""
    from safetensors.numpy import save_file
    from semtools_amd import host

    small_v = 500
    table = synth.table(small_v, seed=5)
    stored = table.astype(np.float16) if dtype == "float16" else np.clip(np.rint(table * 400), -127, 127).astype(np.int8)
    d = tmp_path / "m"
    d.mkdir()
    save_file({"embeddings": stored}, str(d / "model.safetensors"))
    (d / "vocab.txt").write_text("".join(f"w{i}\n" for i in range(small_v - 1)) + "[UNK]\n")
    (d / "config.json").write_text(json.dumps({"normalize": True, "unk_token": "[UNK]"}))
    m = host.StaticModel(gpu_ctx, model_dir=d)
    lines = ["w1 w2 w3 w499 w77", "w5", "", "w400 w401 nothing w3"]
    got = m.encode_with_args(lines, 2048)
    ids, offsets = [], [0]
    for ln in lines:
        ids += [int(w[1:]) for w in ln.split() if w.startswith("w") and w[1:].isdigit() and int(w[1:]) < small_v - 1]
        offsets.append(len(ids))
    want = orc.embed_lines(stored.astype(np.float32), np.array(ids, np.uint32), np.array(offsets, np.uint64), True, 2048)
    assert np.array_equal(np.asarray(got, np.float32).view(np.uint32), want.view(np.uint32))
    m.close()


def test_reference_search_unit_tests_through_the_abi(model, model_dir, tmp_path):
    """The reference's own tests of the path (src/search/mod.rs:252-415), driven through smt_host_search_files
    with the same structure of assertions (their model is the real potion model, ours the synthetic table --
    the asserted properties are the ones the reference asserts)."""
    from semtools_amd import host

    def search(query, files, **kw):
        return json.loads(host.search_files(model, query, files, json=True, **kw))["results"]

    doc1 = tmp_path / "doc1.txt"
    doc1.write_text("w1 w2 w3\nw4 w5 w6\nw7 w8 w9\nw10 w11 w12\nw1 w2 w13\n")
    doc2 = tmp_path / "doc2.txt"
    doc2.write_text("w100 w101\nw1 w2 w3 w4\nw200\n")

    # test_search_documents_basic (:252-274): results sorted by distance ascending
    r = search("w1 w2 w3", [str(doc1)], n_lines=1, top_k=5)
    assert len(r) == 5 and all(r[i]["distance"] <= r[i + 1]["distance"] for i in range(4))
    assert r[0]["match_line_number"] == 0 and r[0]["distance"] < 1e-9

    # test_search_documents_with_max_distance (:276-293): every result strictly under the threshold, top_k ignored
    thr = (r[1]["distance"] + r[2]["distance"]) / 2
    rt = search("w1 w2 w3", [str(doc1)], n_lines=1, top_k=1, max_distance=thr)
    assert len(rt) == 2 and all(x["distance"] < thr for x in rt)
    assert search("w1 w2 w3", [str(doc1)], n_lines=1, top_k=3, max_distance=r[1]["distance"])[-1]["distance"] < r[1]["distance"]

    # test_search_documents_top_k_limit (:295-313)
    assert len(search("w1 w2 w3", [str(doc1)], n_lines=1, top_k=2)) == 2

    # test_search_result_context_calculation (:315-335): n_lines = 1 around line 2 -> start 1, end 4 (exclusive)
    r = search("w7 w8 w9", [str(doc1)], n_lines=1, top_k=1)[0]
    assert (r["match_line_number"], r["start_line_number"], r["end_line_number"]) == (2, 1, 4)

    # test_context_at_file_boundaries (:337-357): clamped to the file
    r = search("w1 w2 w3", [str(doc1)], n_lines=10, top_k=1)[0]
    assert (r["start_line_number"], r["end_line_number"]) == (0, 5)

    # test_multiple_documents_search (:359-379): hits come from both files, best match first
    r = search("w1 w2 w3 w4", [str(doc1), str(doc2)], n_lines=0, top_k=4)
    assert {x["filename"] for x in r} == {str(doc1), str(doc2)} and r[0]["filename"] == str(doc2) and r[0]["distance"] < 1e-9

    # test_empty_documents_handling (:381-391)
    empty = tmp_path / "empty.txt"
    empty.write_text("")
    assert search("w1", [str(empty)], n_lines=1, top_k=3) == []


def test_rows_written_ahead_are_not_durable_until_the_commit(gpu_ctx, tmp_path):
    """smt_sharded_corpus_append_to_file_ex: WRITE_AHEAD puts rows into the file without touching the header -- a reader (a crash
    before the commit) sees the old, consistent prefix; the commit writes what is missing, syncs, and names all rows.  The file ends
    byte-identical to a plain smt_corpus_save of the same rows."""
    import ctypes as C

    import semtools_amd as smt
    from semtools_amd import _lib as L

    emb = synth.unit_rows(5000, seed=61)
    g = smt.Group.from_ctx(gpu_ctx)
    sc = smt.ShardedCorpus(g, empty=True)
    path = tmp_path / "rows.f32"
    fn = L.lib().smt_sharded_corpus_append_to_file_ex
    AHEAD, CREATE = 1, 2
    sc.append(emb[:1200])
    L.check(fn(sc._h, str(path).encode(), 0, 0, AHEAD | CREATE))            # first batch: the file is started (header: 0 rows)
    assert smt.Corpus.load(gpu_ctx, path).rows == 0 and path.stat().st_size == 32 + 1200 * 1024
    sc.append(emb[1200:3000])
    L.check(fn(sc._h, str(path).encode(), 0, 1200, AHEAD))                  # second batch: rows 1200.. only
    assert smt.Corpus.load(gpu_ctx, path).rows == 0
    sc.append(emb[3000:3600])
    L.check(fn(sc._h, str(path).encode(), 0, 3000, 0))                      # the commit writes 3000..3600, syncs, header last
    c = smt.Corpus.load(gpu_ctx, path)
    assert c.rows == 3600 and np.array_equal(c.read_rows(0, 3600), emb[:3600])
    c.close()
    sc.append(emb[3600:])
    L.check(fn(sc._h, str(path).encode(), 3600, 3600, AHEAD))               # everything written ahead ...
    assert smt.Corpus.load(gpu_ctx, path).rows == 3600
    L.check(fn(sc._h, str(path).encode(), 3600, 5000, 0))                   # ... the commit only syncs and names the rows
    plain = smt.Corpus(gpu_ctx)
    plain.append(emb)
    plain.save(tmp_path / "plain.f32")
    assert path.read_bytes() == (tmp_path / "plain.f32").read_bytes()
    with pytest.raises(RuntimeError):                                       # the header must name exactly rows_on_disk rows
        L.check(fn(sc._h, str(path).encode(), 3600, 3600, AHEAD))
    g3 = smt.Group.logical(0, 3)                                            # several shards: no write-ahead (pieces interleave)
    sc3 = smt.ShardedCorpus(g3, rows=emb)
    assert fn(sc3._h, str(tmp_path / "s3.f32").encode(), 0, 0, AHEAD | CREATE) == L.SMT_E_UNSUPPORTED
    sc3.close(); g3.close(); plain.close(); sc.close(); g.close()


def test_a_failed_write_ahead_job_is_still_known_to_the_commit_after_the_corpus_grew(gpu_ctx, tmp_path, monkeypatch):
    """a write-ahead job that fails (ENOSPC, EIO, a copy error -- injected here) leaves a zero-filled hole in the
    file.  Growing the corpus drains the writer (its rows are about to move); that drain used to CLEAR the writer's error, and the
    commit then trusted rows_written and put a header over the hole: embeddings persisted corrupt, no error reported.  Only the
    commit may consume the error; it then writes everything itself.  The file must end byte-identical to a plain save."""
    import semtools_amd as smt
    from semtools_amd import _lib as L

    emb = synth.unit_rows(9000, seed=62)
    g = smt.Group.from_ctx(gpu_ctx)
    sc = smt.ShardedCorpus(g, empty=True)
    path = tmp_path / "rows.f32"
    fn = L.lib().smt_sharded_corpus_append_to_file_ex
    AHEAD, CREATE = 1, 2
    monkeypatch.setenv("SEMTOOLS_DEBUG_FAIL_WRITE_AHEAD", "2")              # the writer's second job fails before it writes a byte
    sc.append(emb[:1200])
    L.check(fn(sc._h, str(path).encode(), 0, 0, AHEAD | CREATE))            # job 1: rows 0..1200, fine
    sc.append(emb[1200:3000])
    L.check(fn(sc._h, str(path).encode(), 0, 1200, AHEAD))                  # job 2: rows 1200..3000 -- fails in the background
    sc.append(emb[3000:9000])                                               # the corpus outgrows its capacity: corpus_reserve drains the writer
    L.check(fn(sc._h, str(path).encode(), 0, 3000, AHEAD))                  # job 3: rows 3000..9000, fine
    L.check(fn(sc._h, str(path).encode(), 0, 9000, 0))                      # the commit: told that everything was written ahead
    c = smt.Corpus.load(gpu_ctx, path)
    got = c.read_rows(0, 9000)
    c.close()
    assert c is not None and got.shape == (9000, 256)
    assert np.array_equal(got[1200:3000], emb[1200:3000]), "the failed job's rows are a hole in the committed file"
    assert np.array_equal(got, emb)
    plain = smt.Corpus(gpu_ctx)
    plain.append(emb)
    plain.save(tmp_path / "plain.f32")
    assert path.read_bytes() == (tmp_path / "plain.f32").read_bytes()
    # the commit consumed the error: the next cycle starts clean (job 4 writes, the commit trusts it)
    monkeypatch.delenv("SEMTOOLS_DEBUG_FAIL_WRITE_AHEAD")
    more = synth.unit_rows(500, seed=63)
    sc.append(more)
    L.check(fn(sc._h, str(path).encode(), 9000, 9000, AHEAD))
    L.check(fn(sc._h, str(path).encode(), 9000, 9500, 0))
    c = smt.Corpus.load(gpu_ctx, path)
    assert c.rows == 9500 and np.array_equal(c.read_rows(9000, 500), more)
    c.close(); plain.close(); sc.close(); g.close()


def test_workspace_written_ahead_equals_the_workspace_written_at_the_end(gpu_ctx, model_dir, prose_files, tmp_path, monkeypatch, capfd):
    """The cold path of a workspace search embeds in batches and writes each batch's rows to line_embeddings.f32 while the next one
    is tokenised (Store::write_rows_ahead; src/workspace/store.rs:402-434 flushes as it goes).  With small batches -- many
    write-ahead calls -- the store on disk and the answers are byte-identical to SEMTOOLS_WRITE_AHEAD=0, also when a second call
    appends a new file to the existing store."""
    from semtools_amd import host

    files = []
    lines = prose_files[1][1]
    for i in range(6):
        f = tmp_path / f"doc{i}.txt"
        f.write_text("\n".join(lines[i * 40:(i + 1) * 40 + 100]) + "\n")
        files.append(str(f))
    monkeypatch.setenv("SEMTOOLS_EMBED_BATCH", "64")
    results, stores = {}, {}
    for mode in ("1", "0"):
        home = tmp_path / f"home{mode}"
        home.mkdir()
        monkeypatch.setenv("HOME", str(home))
        monkeypatch.setenv("SEMTOOLS_WRITE_AHEAD", mode)
        m = host.StaticModel(gpu_ctx, model_dir=model_dir[0])
        try:
            host.workspace_use(None, "wa")
            r1 = host.search_with_workspace(m, lines[17], files[:4], workspace_name="wa", n_lines=1, top_k=5)
            r2 = host.search_with_workspace(m, lines[17], files, workspace_name="wa", n_lines=1, top_k=5)   # two more files: an append
            from semtools_amd import _lib as L
            L.lib().smt_host_timing_json.restype = ctypes.c_void_p
            ptr = L.lib().smt_host_timing_json()
            timing = json.loads(ctypes.string_at(ptr).decode()) if ptr else {}
        finally:
            m.close()
        root = home / ".semtools" / "workspaces" / "wa"
        results[mode] = (r1, r2)
        stores[mode] = ((root / "line_embeddings.f32").read_bytes(), json.loads((root / "line_rows.json").read_text())["extents"])
        assert not (root / "line_embeddings.f32.tmp").exists()
        if mode == "1" and timing:
            assert any("rows_written_ahead" in k for k in timing), timing
    capfd.readouterr()
    assert results["1"] == results["0"]
    assert stores["1"][0] == stores["0"][0]
    strip = lambda ext: [(os.path.basename(e["path"]), e["first_row"], e["n_rows"]) for e in ext]
    assert strip(stores["1"][1]) == strip(stores["0"][1])


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_workspace_random_life_of_a_store(model, model_dir, tmp_path, monkeypatch, capfd, seed):
    """A workspace lived in for 120 random steps -- new files, files that grow, shrink or are rewritten, deleted files and prune,
    searches over random subsets with and without a threshold -- against a model of it: after every search the hits are exactly
    what Store::search_line_embeddings (src/workspace/store.rs:481-546, oracle restatement) returns on the CURRENT contents of the
    searched files: as many hits, the same sorted distances (1e-5), every hit an eligible (file, line) with that distance.  One GPU
    and three shards; the store sees holes, re-used extents, documents that move, and a second process state (status) in between."""
    from semtools_amd import host

    monkeypatch.setenv("HOME", str(tmp_path))
    monkeypatch.delenv("SEMTOOLS_WORKSPACE", raising=False)
    table = model_dir[1]
    rng = np.random.default_rng(1000 + seed)
    ws = f"life{seed}"
    host.workspace_use(None, ws)
    pool = synth.pseudo_prose(3000, vocab_size=V - 1, seed=50 + seed)
    pool[7] = ""                                  # zero vectors take part
    pool[8] = "unknownword"
    files = {}                                    # path -> lines now on disk
    clock = [1_800_000_000]
    emb_of = {}

    def emb(line):
        if line not in emb_of:
            emb_of[line] = oracle_embed(table, [line], 2048)[0]
        return emb_of[line]

    def write(path, lines):
        with open(path, "w") as f:
            f.write("".join(l + "\n" for l in lines))
        clock[0] += 10
        os.utime(path, (clock[0], clock[0]))
        files[path] = lines

    def some_lines(lo, hi):
        return [pool[int(i)] for i in rng.integers(0, len(pool), size=int(rng.integers(lo, hi + 1)))]

    n_made = 0
    searched = 0
    for step in range(120):
        op = rng.choice(["new", "new", "grow", "shrink", "rewrite", "delete", "search", "search", "search"]) if files else "new"
        if op == "new":
            p = str(tmp_path / f"f{seed}_{n_made}.txt")
            n_made += 1
            write(p, some_lines(1, 300))
            continue
        victim = sorted(files)[int(rng.integers(0, len(files)))]
        if op == "grow":
            write(victim, files[victim] + some_lines(1, 200))
        elif op == "shrink":
            write(victim, files[victim][:max(1, len(files[victim]) // 2)])
        elif op == "rewrite":
            write(victim, some_lines(1, 300))
        elif op == "delete":
            os.unlink(victim)
            del files[victim]
            text = host.workspace_prune(model.ctx, ws)
            assert "stale" in text.lower() or "clean" in text.lower()
        else:
            names = sorted(files)
            want = [names[int(i)] for i in rng.choice(len(names), size=int(rng.integers(1, len(names) + 1)), replace=False)]
            k = int(rng.integers(1, 25))
            thr = None if rng.random() < 0.5 else float(rng.uniform(0.6, 1.0))
            query = pool[int(rng.integers(10, len(pool)))]
            js = json.loads(host.search_with_workspace(model, query, want, workspace_name=ws, n_lines=0, top_k=k, max_distance=thr, json=True))
            capfd.readouterr()
            e = np.stack([emb(l) for p in want for l in files[p]])
            path = np.concatenate([np.full(len(files[p]), i, np.uint32) for i, p in enumerate(want)])
            line = np.concatenate([np.arange(len(files[p]), dtype=np.int32) for p in want])
            q = oracle_embed(table, [query], 512)[0]
            exp = orc.search_line_embeddings(e, path, line, q, np.arange(len(want), dtype=np.uint32), k, thr)
            got = js["results"]
            note = (seed, step, len(want), k, thr)
            assert len(got) == len(exp), note
            np.testing.assert_allclose([r["distance"] for r in got], [r["distance"] for r in exp], rtol=0, atol=1e-5, err_msg=str(note))
            seen = set()
            for r in got:
                key = (r["filename"], r["match_line_number"])
                assert r["filename"] in want and 0 <= r["match_line_number"] < len(files[r["filename"]]) and key not in seen, (note, r)
                seen.add(key)
                ln = files[r["filename"]][r["match_line_number"]]
                assert r["content"] == ln, (note, r)
                assert abs(orc.cosine(q, emb(ln)) - r["distance"]) < 1e-5, (note, r)
            searched += 1
            if step % 7 == 0:
                st = json.loads(host.workspace_status(model.ctx, ws, json=True))
                assert st["total_documents"] >= len(want)
    assert searched >= 15
