"""CPU: host-layer pieces that need no GPU -- output formatting primitives (Rust Display / serde_json
number formats), str::lines / to_lowercase, `workspace use`, CLI argument errors -- and the test-side
format restatement (tests/refimpl.py) agreeing with the library's."""
import json
import os
import subprocess

import numpy as np
import pytest

from semtools_amd import _lib as L, host
from tests import refimpl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "semtools_amd", "bin", "semtools")


def test_host_symbols_exported():
    lib = L.lib()
    src = open(os.path.join(ROOT, "include", "semtools_host.h")).read()
    import re
    declared = sorted(set(re.findall(r"\b(smt_host_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", src, flags=re.S))))
    assert declared == sorted(L.HOST_EXPORTS)
    for n in declared:
        assert hasattr(lib, n), n


@pytest.mark.parametrize("v,disp,js", [
    (0.1, "0.1", "0.1"), (1e-7, "0.0000001", "1e-7"), (1.0, "1", "1.0"), (0.0, "0", "0.0"),
    (0.7147720518571459, "0.7147720518571459", "0.7147720518571459"), (1234.5, "1234.5", "1234.5"),
    (1e16, "10000000000000000", "1e16"), (0.00001, "0.00001", "0.00001"), (2.5e-6, "0.0000025", "2.5e-6"),
    (123456789012345680000.0, "123456789012345680000", "1.2345678901234568e20"), (5e-324, None, "5e-324"),
    (100.0, "100", "100.0"), (1.1102230246251565e-16, "0.00000000000000011102230246251565", "1.1102230246251565e-16"),
])
def test_number_formats_match_rust_and_serde(v, disp, js):
    if disp is not None:
        assert host.format_float(v, 0) == disp == refimpl.rust_f64(v)
    assert host.format_float(v, 2) == js == refimpl.serde_f64(v)


def test_f32_display_is_shortest_f32():
    for v in (0.1, 0.71477205, 1.0 / 3.0, 1e-7, 16777216.0, 0.30000001192092896):
        assert host.format_float(v, 1) == refimpl.rust_f32(v)
    assert host.format_float(0.1, 1) == "0.1" and host.format_float(1.0 / 3.0, 1) == "0.33333334"


def test_random_floats_roundtrip_and_agree():
    rng = np.random.default_rng(0)
    for x in np.concatenate([rng.random(200), rng.random(50) * 1e-6, 1.0 - rng.random(50) * 1e-9]):
        d, j = host.format_float(x, 0), host.format_float(x, 2)
        assert float(d) == x and float(j) == x
        assert d == refimpl.rust_f64(x) and j == refimpl.serde_f64(x)


@pytest.mark.parametrize("content", ["", "a", "a\n", "a\nb", "a\r\nb\r\n", "\n", "\n\n", "a\n\nb\n", "a\rb\n", "x\r", "é\nü\n"])
def test_lines_like_rust(content):
    assert host.split_lines(content) == refimpl.rust_lines(content)


@pytest.mark.parametrize("tail", ["", "\n", "\r", "\r\n", "last"])
def test_lines_of_a_big_content_cut_into_slices(tail):
    """Contents of 4 MiB and more have their line ends found on several threads, slice by slice: every mix of "\\n", "\\r\\n", lone
    "\\r" and empty lines -- also right at a slice border -- must come out as str::lines() gives it (mod.rs:51)."""
    rng = np.random.default_rng(5)
    pieces = ["word", "two words", "", "x", "carriage\rinside", "é ü"]
    ends = ["\n", "\r\n", "\n", "\n\n", "\r\r\n"]
    parts = []
    size = 0
    while size < (5 << 20):
        p = pieces[int(rng.integers(len(pieces)))] * int(rng.integers(1, 30)) + ends[int(rng.integers(len(ends)))]
        parts.append(p)
        size += len(p)
    content = "".join(parts) + tail
    assert len(content.encode()) >= (4 << 20)
    assert host.split_lines(content) == refimpl.rust_lines(content)


def test_rust_lines_examples():
    assert refimpl.rust_lines("Line 1\nLine 2\nLine 3") == ["Line 1", "Line 2", "Line 3"]   # mod.rs:419-430
    assert refimpl.rust_lines("") == []                                                       # mod.rs:434-441
    assert refimpl.rust_lines("a\r\nb") == ["a", "b"] and refimpl.rust_lines("\n") == [""]


def test_to_lowercase():
    assert host.to_lowercase("Hello World") == "hello world"
    assert host.to_lowercase("GOODBYE wOrLd 123 ÄÖÜ Ж") == "goodbye world 123 äöü ж"


def test_to_lowercase_is_rusts_not_the_locales():
    """Rust's str::to_lowercase (src/search/mod.rs:63) = the Unicode lower-case mapping incl. the multi-code-point
    case (U+0130) and Final_Sigma -- which is what Python's str.lower() implements too.  No locale involved."""
    import random

    cases = ["İstanbul", "ΟΔΥΣΣΕΥΣ", "ΣΑΣ ΣΟΦΟΣ.", "ΑΣ", "Σ", "aΣ", "AΣ'", "Σa", "ΧΑΟΣ-ΧΑΟΣ ΧΑΟΣ", "ǅ ǈ ǋ ǲ", "ẞ STRASSE",
             "ԱԲԳ ႠႡႢ ᲐᲑᲒ", "ＦＵＬＬ Ｗｉｄｔｈ", "Ⅻ Ⓐ Ꙁ Ⰰ", "𐐀𐐁 𞤀𞤁", "ÀÉÎÕÜ ĀĂĄ ŁŃŚ ƁƂƄ ǍǏ", "ΆΈΉΊΌΎΏ ΪΫ ϏϘϚ", "ЀЁЂ АБВ ЯѠ ҐҒ ӁӃ ԀԂ",
             "mixed ASCII and ÜNÏCÖDÉ 123 !?", "\udcff".encode("utf-8", "surrogatepass").decode("utf-8", "replace")]
    rng = random.Random(7)
    pool = [c for c in (chr(cp) for cp in list(range(0x20, 0x250)) + list(range(0x370, 0x530)) + list(range(0x1E00, 0x2000))
                        + [0x3A3, 0x3A3, 0x27, 0x2E, 0x300, 0x301, 0xAD]) if c.isprintable() or c in "\u0300\u0301\u00ad"]
    cases += ["".join(rng.choice(pool) for _ in range(rng.randint(1, 40))) for _ in range(400)]
    for s in cases:
        assert host.to_lowercase(s) == s.lower(), repr(s)


def test_workspace_use_writes_config_and_prints_reference_text(tmp_path, monkeypatch):
    monkeypatch.setenv("HOME", str(tmp_path))
    text = host.workspace_use(None, "proj")
    assert text == ("Workspace 'proj' configured.\nTo activate it, run:\n  export SEMTOOLS_WORKSPACE=proj\n\n"
                    "Or add this to your shell profile (.bashrc, .zshrc, etc.)\n\n"
                    "Or use the `--workspace` option on the commands that support it\n")   # src/cmds/workspace.rs:46-53
    cfg = json.load(open(tmp_path / ".semtools" / "workspaces" / "proj" / "config.json"))
    assert cfg == {"name": "proj", "root_dir": str(tmp_path / ".semtools" / "workspaces" / "proj"),
                   "in_batch_size": 5000, "oversample_factor": 3}                              # src/workspace/mod.rs:15-26
    assert list(cfg) == ["name", "root_dir", "in_batch_size", "oversample_factor"]


def test_cli_binary_argument_handling(tmp_path):
    env = dict(os.environ, HOME=str(tmp_path))
    r = subprocess.run([CLI], capture_output=True, text=True, env=env)
    assert r.returncode == 2 and "search" in r.stderr and "workspace" in r.stderr
    r = subprocess.run([CLI, "workspace", "use", "w1"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.startswith("Workspace 'w1' configured.")
    assert (tmp_path / ".semtools" / "workspaces" / "w1" / "config.json").exists()


def test_refimpl_json_layout_matches_serde_pretty():
    res = [dict(filename="a \"q\".txt", lines=["x", "y\tz"], start=0, end=2, match_line=1, distance=0.25)]
    txt = refimpl.search_results_json(res)
    assert json.loads(txt)["results"][0]["content"] == "x\ny\tz"
    assert txt.splitlines()[0] == "{" and txt.splitlines()[1] == '  "results": [' and '"distance": 0.25,' in txt
    assert refimpl.search_results_json([]) == '{\n  "results": []\n}\n'
