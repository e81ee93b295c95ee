"""CPU: the C-ABI library loads, exports every symbol include/semtools_hip.h declares, its host-only
helpers agree with the oracle, and every compute entry point FAILS LOUDLY without a GPU
(there is no CPU fallback to fall into)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import semtools_amd as smt
from semtools_amd import _lib as L
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "semtools_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smt_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built_in_tree():
    assert os.path.exists(L.LIB_PATH), "run semtools_amd/csrc/build.sh (or __graft_entry__.build())"
    assert os.path.dirname(L.LIB_PATH).startswith(ROOT)


def test_every_declared_symbol_is_exported():
    names = _declared()
    assert len(names) >= 30
    lib = L.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in semtools_hip.h but not exported"
    assert sorted(L.EXPORTS) == names, "semtools_amd/_lib.py EXPORTS out of sync with the header"
    dyn = subprocess.check_output(["nm", "-D", "--defined-only", L.LIB_PATH], text=True)
    for n in names:
        assert re.search(rf"\bT {n}\b", dyn), n


def test_contains_gfx950_code_object():
    blob = open(L.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"scan_topk_kernel" in blob


def test_host_ids_match_oracle():
    lib = L.lib()
    for path, line in (("/test/doc1.txt", 0), ("a", 7), ("päth/ü.txt", 123456), ("x", -1)):
        assert lib.smt_line_embedding_id(path.encode(), line) == orc.line_embedding_id(path, line)
        assert lib.smt_doc_meta_id(path.encode()) == orc.doc_meta_id(path)
    assert lib.smt_fnv1a_hash(b"foobar", 6) == 0x85944171F73967E8


def test_host_merge_topk():
    rng = np.random.default_rng(0)
    n_lists, nq, k = 5, 3, 6
    rows = np.full((n_lists, nq, k), np.iinfo(np.uint64).max, np.uint64)
    dist = np.full((n_lists, nq, k), np.inf)
    truth = [[] for _ in range(nq)]
    for l in range(n_lists):
        for q in range(nq):
            n = int(rng.integers(0, k + 1))
            d = np.sort(np.round(rng.random(n), 1))                 # coarse -> ties across lists
            r = np.sort(rng.choice(1000, n, replace=False)) + 1000 * l
            order = np.lexsort((r, d))
            rows[l, q, :n], dist[l, q, :n] = r[order], d[order]
            truth[q] += list(zip(d[order].tolist(), (r[order]).tolist()))
    mr, md, cnt = smt.merge_topk(rows, dist, 4)
    for q in range(nq):
        want = sorted(truth[q])[:4]
        assert cnt[q] == len(want)
        assert mr[q, : len(want)].tolist() == [w[1] for w in want]
        assert md[q, : len(want)].tolist() == [w[0] for w in want]
        assert (mr[q, len(want):] == np.iinfo(np.uint64).max).all() and np.isinf(md[q, len(want):]).all()


def test_a_cxx_exception_becomes_an_error_code_not_an_unwinding_abi():
    """include/semtools_hip.h: "Nothing unwinds or aborts across the ABI."  Every int-returning entry point is a function-try-block
    (csrc/common.h api_catch).  Provoked here without a GPU: smt_merge_topk gathers its candidates into a host vector; with the
    process' address space capped just above what it already uses, that vector cannot grow -- std::bad_alloc inside the library --
    and the call must come back with SMT_E_NOMEM and a message, the process alive."""
    code = r'''
import ctypes as C, resource, sys
import numpy as np
sys.path.insert(0, %r)
from semtools_amd import _lib as L
lib = L.lib()
k_in = 8 << 20
rows = np.arange(k_in, dtype=np.uint64)
dist = np.linspace(0.0, 1.0, k_in)
out_r = np.empty(4, np.uint64); out_d = np.empty(4, np.float64); cnt = np.empty(1, np.uint64)
used = int(open("/proc/self/statm").read().split()[0]) * resource.getpagesize()
resource.setrlimit(resource.RLIMIT_AS, (used + (48 << 20), resource.RLIM_INFINITY))     # the candidates need 128 MiB and more
rc = lib.smt_merge_topk(L.np_ptr(rows), L.np_ptr(dist), 1, 1, k_in, 4, L.np_ptr(out_r), L.np_ptr(out_d), L.np_ptr(cnt))
print("rc", rc, "|", lib.smt_last_error().decode())
''' % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.startswith("rc -3 |") and "memory" in p.stdout, p.stdout     # SMT_E_NOMEM


@pytest.mark.skipif(L.lib().smt_device_count() > 0, reason="a GPU is present")
def test_no_gpu_means_loud_failure_not_fallback():
    with pytest.raises(smt.SmtError) as e:
        smt.Context(0)
    assert e.value.code == L.SMT_E_HIP
    h = C.c_void_p()
    assert L.lib().smt_ctx_create(0, C.byref(h)) == L.SMT_E_HIP and not h
    assert b"hip" in L.lib().smt_last_error().lower()


def test_null_handles_are_rejected():
    lib = L.lib()
    assert lib.smt_corpus_rows(None) == 0 and lib.smt_corpus_dim(None) == 0
    assert lib.smt_search(None, None, 1, 1, float("nan"), 0, None, 0, 0, None, None, None, 0) == L.SMT_E_INVALID
    assert lib.smt_embed(None, None, None, 0, 0, None, None, None) == L.SMT_E_INVALID
    assert lib.smt_ctx_synchronize(None) == L.SMT_E_INVALID
    lib.smt_ctx_destroy(None); lib.smt_model_destroy(None); lib.smt_corpus_destroy(None)  # no-ops


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "semtools_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".sh")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.replace("the oracle", "").replace("oracle/", "ORACLE_DOC/").lower() or \
                    all("import" not in ln and "#include" not in ln and "dlopen" not in ln and "CDLL" not in ln
                        for ln in text.splitlines() if "oracle" in ln.lower()), f


def test_every_int_entry_point_is_a_function_try_block():
    """common.h / INTEGRATION.md: no C++ exception unwinds into a C, Rust or ctypes caller -- every int-returning extern "C" entry
    point is a function-try-block ending in api_catch() (ADVICE r4 found smt_ctx_create without one)."""
    import glob
    import re

    root = os.path.join(ROOT, "semtools_amd", "csrc")
    missing, seen = [], 0
    # (include/semtools_hip.h's entry points; the host layer of semtools_host.h catches inside its bodies -- host_capi.cpp `fail(e)`)
    for path in sorted(glob.glob(os.path.join(root, "*.cpp")) + glob.glob(os.path.join(root, "*.hip"))):
        text = open(path).read()
        for m in re.finditer(r"^int (smt_\w+)\(", text, re.M):
            depth, i = 1, m.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            rest = text[i:i + 40].lstrip()
            if rest.startswith(";"):
                continue                      # a declaration
            seen += 1
            if not rest.startswith("try"):
                missing.append(f"{os.path.basename(path)}: {m.group(1)}")
    assert seen > 60 and not missing, (seen, missing)


def test_the_rccl_double_exports_what_the_library_binds():
    """tests/fake_rccl stands in for librccl.so.1 in the multi-process GPU tests (tests/test_gpu_spmd.py): it must carry RCCL's
    SONAME -- load_rccl's dlopen(RTLD_NOLOAD) finds it by that -- and every nccl* symbol csrc/group.cpp resolves."""
    import re
    import subprocess

    from tests import fake_rccl

    lib = fake_rccl.build()
    src = open(os.path.join(ROOT, "semtools_amd", "csrc", "group.cpp")).read()
    bound = {"nccl" + m for m in re.findall(r"^\s*SMT_RCCL_SYM\((\w+)\)", src, flags=re.M)}
    assert len(bound) >= 12
    syms = subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True)
    have = {line.split()[-1] for line in syms.splitlines() if line.strip()}
    assert bound <= have, sorted(bound - have)
    dyn = subprocess.check_output(["readelf", "-d", lib], text=True)
    assert "librccl.so.1" in dyn
    # ... and the product neither links nor names it
    for root, _, files in os.walk(os.path.join(ROOT, "semtools_amd")):
        for f in files:
            if f.endswith((".cpp", ".hip", ".h", ".py", ".sh")):
                assert "fake_rccl" not in open(os.path.join(root, f), errors="replace").read(), f
