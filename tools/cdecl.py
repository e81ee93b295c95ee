"""Minimal parser for the function declarations of include/semtools_hip.h (plain C, one prototype per statement).
Used by tools/gen_rust_ffi.py (writes rust/src/search/hip_ffi.rs) and by tests/test_rust_ffi.py (checks the
committed Rust extern block against the header)."""
import re

_RUST_SCALAR = {"int": "c_int", "uint64_t": "u64", "uint32_t": "u32", "int32_t": "i32", "int64_t": "i64",
                "uint8_t": "u8", "double": "f64", "float": "f32", "char": "c_char", "void": "c_void"}


def _camel(name):
    return "".join(p.capitalize() for p in name.split("_"))


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def c_type_to_rust(ctype):
    """'const float *const *' -> '*const *const f32'; 'smt_ctx **' -> '*mut *mut SmtCtx'; 'uint32_t' -> 'u32'."""
    toks = re.findall(r"[A-Za-z_][A-Za-z0-9_]*|\*", ctype)
    toks = [t for t in toks if t != "struct"]
    # base type = first identifier that is not 'const'
    base_i = next(i for i, t in enumerate(toks) if t not in ("const", "*"))
    base = toks[base_i]
    base_const = "const" in toks[:base_i] or (base_i + 1 < len(toks) and toks[base_i + 1] == "const")
    rust = _RUST_SCALAR.get(base) or _camel(base)
    # walk the declarator: every '*' adds a pointer level; a 'const' right after a '*' qualifies THAT pointer
    rest = toks[base_i + 1:]
    if rest and rest[0] == "const":
        rest = rest[1:]
    pointee_const = base_const
    out = rust
    i = 0
    while i < len(rest):
        assert rest[i] == "*", ctype
        out = ("*const " if pointee_const else "*mut ") + out
        pointee_const = i + 1 < len(rest) and rest[i + 1] == "const"
        i += 2 if pointee_const else 1
    return out


def parse_header(path):
    """-> list of (name, return_ctype, [(param_ctype, param_name)])."""
    src = strip_comments(open(path).read())
    src = re.sub(r"#[^\n]*", " ", src)                       # preprocessor lines
    src = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    src = re.sub(r"typedef[^;]*;", " ", src)
    src = src.replace('extern "C" {', " ").replace("}", " ")
    out = []
    for stmt in src.split(";"):
        stmt = " ".join(stmt.split())
        m = re.match(r"^(.*?)\b(smt_[a-z0-9_]+)\s*\((.*)\)$", stmt)
        if not m:
            continue
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        plist = []
        if params and params != "void":
            for p in params.split(","):
                p = p.strip()
                pm = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", p)
                plist.append((pm.group(1).strip(), pm.group(2)))
        out.append((name, ret, plist))
    return out


def parse_rust_externs(path):
    """-> dict name -> (return_rust_type or None, [rust types]) for every `pub fn` inside extern "C" blocks."""
    src = re.sub(r"//[^\n]*", " ", open(path).read())
    out = {}
    for block in re.findall(r'extern\s+"C"\s*\{(.*?)\n\}', src, flags=re.S):
        for m in re.finditer(r"pub\s+fn\s+(smt_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
            name, params, ret = m.group(1), " ".join(m.group(2).split()), m.group(3)
            types = []
            if params:
                for p in params.split(","):
                    p = p.strip()
                    if p:
                        types.append(" ".join(p.split(":", 1)[1].split()))
            out[name] = (" ".join(ret.split()) if ret else None, types)
    return out
