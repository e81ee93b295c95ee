"""Test-side restatement of the reference's OUTPUT layer (src/cmds/search.rs:35-110, :23-32, :208-241;
src/json_mode.rs) and of its string handling (str::lines, f64/f32 Display, serde_json pretty).
Numbers come from the oracle; this module only formats them the way the Rust code does."""
import json as _json

import numpy as np


def rust_lines(content):
    """str::lines(): split on \\n, strip one trailing \\r of a terminated line, no trailing empty piece."""
    if content == "":
        return []
    parts = content.split("\n")
    last_terminated = content.endswith("\n")
    if last_terminated:
        parts = parts[:-1]
    out = []
    for i, p in enumerate(parts):
        terminated = i < len(parts) - 1 or last_terminated
        out.append(p[:-1] if terminated and p.endswith("\r") else p)
    return out


def rust_f64(x):
    return np.format_float_positional(np.float64(x), unique=True, trim="-")


def rust_f32(x):
    return np.format_float_positional(np.float32(x), unique=True, trim="-")


def serde_f64(x):
    """ryu: decimal when -5 < kk <= 16 else exponent form, always with '.0' on integral decimals."""
    x = float(x)
    if x == 0:
        return "0.0"
    s = np.format_float_scientific(np.float64(abs(x)), unique=True, trim="-", exp_digits=1)  # d.ddde[+-]x
    mant, exp = s.split("e")
    digits = mant.replace(".", "")
    kk = int(exp) + 1
    n = len(digits)
    sign = "-" if x < 0 else ""
    if n <= kk <= 16:
        return sign + digits + "0" * (kk - n) + ".0"
    if 0 < kk <= 16:
        return sign + digits[:kk] + "." + digits[kk:]
    if -5 < kk <= 0:
        return sign + "0." + "0" * (-kk) + digits
    return sign + digits[0] + ("." + digits[1:] if n > 1 else "") + "e" + str(kk - 1)


def print_search_results(results, is_tty=False):
    """results: list of dict(filename, lines, start, end, match_line, distance)."""
    out = []
    for r in results:
        out.append(f"{r['filename']}:{r['start']}::{r['end']} ({rust_f64(r['distance'])})")
        for i, line in enumerate(r["lines"]):
            ln = r["start"] + i
            body = f"{ln + 1:4}: {line}"
            out.append(f"\x1b[43m\x1b[30m{body}\x1b[0m" if is_tty and ln == r["match_line"] else body)
        out.append("")
    return "".join(s + "\n" for s in out)


def _json_str(s):
    return _json.dumps(s, ensure_ascii=False)


def search_results_json(results):
    if not results:
        return '{\n  "results": []\n}\n'
    items = []
    for r in results:
        content = "\n".join(r["lines"]) if "lines" in r else r["content"]
        items.append("    {\n"
                     f"      \"filename\": {_json_str(r['filename'])},\n"
                     f"      \"start_line_number\": {r['start']},\n"
                     f"      \"end_line_number\": {r['end']},\n"
                     f"      \"match_line_number\": {r['match_line']},\n"
                     f"      \"distance\": {serde_f64(r['distance'])},\n"
                     f"      \"content\": {_json_str(content)}\n"
                     "    }")
    return "{\n  \"results\": [\n" + ",\n".join(items) + "\n  ]\n}\n"


def results_from_oracle(docs, oracle_results):
    """docs: list of (filename, lines); oracle_results: dicts(doc, match_line, start, end, distance)."""
    out = []
    for r in oracle_results:
        fn, lines = docs[r["doc"]]
        out.append(dict(filename=fn, lines=lines[r["start"]:r["end"]], start=r["start"], end=r["end"],
                        match_line=r["match_line"], distance=r["distance"]))
    return out
