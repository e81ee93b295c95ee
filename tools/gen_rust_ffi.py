#!/usr/bin/env python3
"""Writes rust/src/search/hip_ffi.rs: the complete `extern "C"` block for include/semtools_hip.h (every exported
function, same order, same argument names).  Run after changing the header: python tools/gen_rust_ffi.py"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cdecl  # noqa: E402

HEADER = os.path.join(ROOT, "include", "semtools_hip.h")
OUT = os.path.join(ROOT, "rust", "src", "search", "hip_ffi.rs")

PRELUDE = '''//! FFI declarations for libsemtools_hip.so -- GENERATED from include/semtools_hip.h by tools/gen_rust_ffi.py,
//! do not edit by hand (tests/test_rust_ffi.py compares this file with the header: names, arity, scalar types).
//!
//! Uncompiled in this repository: the build container has no Rust toolchain.  A semtools maintainer adds this
//! file as `src/search/hip_ffi.rs`, the wrappers next to it (`hip.rs`, `../workspace/hip_store.rs`) and
//! `build.rs`; see INTEGRATION.md.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

macro_rules! opaque { ($($name:ident),*) => { $( #[repr(C)] pub struct $name { _private: [u8; 0] } )* } }
opaque!(SmtCtx, SmtModel, SmtCorpus, SmtIvfpq, SmtGroup, SmtShardedCorpus, SmtShardedIvfpq, SmtShardedModel);

/// half-open range of corpus rows [begin, end)
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct SmtRange {
    pub begin: u64,
    pub end: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct SmtIvfpqParams {
    pub nlist: u32,
    pub m: u32,
    pub nbits: u32,
    pub train_iters: u32,
    pub train_sample: u64,
    pub reserved: u32,
    pub local_pca: u32,
}

'''

POSTLUDE = '''
/// `anyhow` error carrying the library's thread-local message (the reference's error type on this path).
pub fn check(rc: c_int) -> anyhow::Result<()> {
    if rc == SMT_OK {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(smt_last_error()) }.to_string_lossy().into_owned();
    anyhow::bail!("semtools_hip error {rc}: {msg}")
}
'''

RESERVED = {"type", "ref", "mod", "fn", "in", "box", "move", "match", "loop", "where", "use", "impl", "self"}


def main():
    src = open(HEADER).read()
    consts = re.findall(r"#define\s+(SMT_[A-Z_0-9]+)\s+\(?(-?\d+)u?\)?", src)
    lines = [PRELUDE]
    for name, val in consts:
        ty = "u32" if name in ("SMT_DIM",) else ("usize" if name == "SMT_UNIQUE_ID_BYTES" else "c_int")
        lines.append(f"pub const {name}: {ty} = {val};\n")
    lines.append('\nextern "C" {\n')
    for name, ret, params in cdecl.parse_header(HEADER):
        args = []
        for ctype, pname in params:
            if pname in RESERVED:
                pname += "_"
            args.append(f"{pname}: {cdecl.c_type_to_rust(ctype)}")
        r = "" if ret == "void" else f" -> {cdecl.c_type_to_rust(ret)}"
        one = f"    pub fn {name}({', '.join(args)}){r};\n"
        if len(one) > 118:
            one = f"    pub fn {name}(\n" + "".join(f"        {a},\n" for a in args) + f"    ){r};\n"
        lines.append(one)
    lines.append("}\n")
    lines.append(POSTLUDE)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write("".join(lines))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
