"""CPU: the committed Rust `extern "C"` block (rust/src/search/hip_ffi.rs) agrees with include/semtools_hip.h --
every exported function is declared, with the same number of arguments and the same scalar / pointer types.
The Rust side cannot be compiled here (no toolchain), so this is the mechanical check that keeps it honest."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cdecl  # noqa: E402

HEADER = os.path.join(ROOT, "include", "semtools_hip.h")
FFI = os.path.join(ROOT, "rust", "src", "search", "hip_ffi.rs")


def test_extern_block_matches_the_header():
    decls = cdecl.parse_header(HEADER)
    rust = cdecl.parse_rust_externs(FFI)
    assert len(decls) >= 60
    assert sorted(n for n, _, _ in decls) == sorted(rust), "function sets differ (run tools/gen_rust_ffi.py)"
    for name, ret, params in decls:
        r_ret, r_params = rust[name]
        assert len(r_params) == len(params), f"{name}: arity {len(r_params)} vs {len(params)}"
        want_ret = None if ret == "void" else cdecl.c_type_to_rust(ret)
        assert r_ret == want_ret, f"{name}: return {r_ret} vs {want_ret}"
        for (ctype, pname), rtype in zip(params, r_params):
            assert rtype == cdecl.c_type_to_rust(ctype), f"{name}({pname}): {rtype} vs {ctype}"


def test_type_mapping_spot_checks():
    m = cdecl.c_type_to_rust
    assert m("uint32_t") == "u32" and m("double") == "f64" and m("int") == "c_int"
    assert m("const float *") == "*const f32" and m("uint64_t *") == "*mut u64"
    assert m("smt_ctx **") == "*mut *mut SmtCtx" and m("const smt_corpus *") == "*const SmtCorpus"
    assert m("const float *const *") == "*const *const f32" and m("uint64_t *const *") == "*const *mut u64"
    assert m("const char *") == "*const c_char" and m("void **") == "*mut *mut c_void"


def test_wrappers_only_call_declared_functions():
    rust = cdecl.parse_rust_externs(FFI)
    for rel in ("rust/src/search/hip.rs", "rust/src/workspace/hip_store.rs"):
        src = open(os.path.join(ROOT, rel)).read()
        for name in set(re.findall(r"\b(smt_[a-z0-9_]+)\s*\(", src)):
            assert name in rust, f"{rel} calls {name}, which hip_ffi.rs does not declare"


def test_constants_match():
    hdr = open(HEADER).read()
    ffi = open(FFI).read()
    for name, val in re.findall(r"#define\s+(SMT_[A-Z_0-9]+)\s+\(?(-?\d+)u?\)?", hdr):
        assert re.search(rf"pub const {name}: \w+ = {val};", ffi), name
