"""CPU: the native tokenizer.json reader (semtools_amd/csrc/host/hf_tokenizer.cpp) against the Hugging Face
`tokenizers` wheel -- the SAME Rust crate the reference reaches through model2vec-rs (Cargo.toml:36) -- so, unlike
the embed / cosine arithmetic, this step of the path has a real anchor in this container.  Tokenizers are trained
here (no network), then every id sequence must be identical on multilingual, accented, CJK, punctuation-heavy,
control-character and special-token-bearing text."""
import ctypes as C
import random

import numpy as np
import pytest

from semtools_amd import _lib as L

tokenizers = pytest.importorskip("tokenizers")
from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, trainers  # noqa: E402

CORPUS = [
    "the quick brown fox jumps over the lazy dog, again and again!",
    "Semantic search over plain-text files: embeddings, cosine distance, top-k.",
    "Ünïcödé straße façade naïve coöperate résumé São Paulo Ångström",
    "中文字符 混合 text 日本語のテキスト 한국어 텍스트 and English",
    "Ελληνικά κείμενα ΟΔΥΣΣΕΥΣ Русский текст Україна İstanbul",
    "numbers 12345 67.89 1e-5 0xDEADBEEF; symbols $ + < = > ^ ` | ~ and punct !?.,:;()[]{}",
    "tabs\tand\nnewlines\r\nand  double  spaces   everywhere",
    "emoji 🙂🙃 mixed 👍🏽 with text, zero​width and soft­hyphen",
] * 6


def native(path):
    h = C.c_void_p()
    L.check(L.lib().smt_host_tokenizer_load(str(path).encode(), C.byref(h)))

    def enc(text):
        cap = 4 * len(text.encode()) + 16
        ids = np.empty(cap, np.uint32)
        n = C.c_uint64()
        L.check(L.lib().smt_host_tokenizer_encode(h, text.encode("utf-8", "surrogatepass").decode("utf-8", "replace").encode(),
                                                  L.np_ptr(ids), cap, C.byref(n)))
        return ids[: n.value].tolist()

    enc.handle = h
    return enc


def samples(seed, n=300):
    rng = random.Random(seed)
    pool = (list("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789") + list(" \t\n.,;:!?()[]{}'\"-_/\\@#$%^&*+=<>|~`")
            + list("äöüßéèêëçñøåÄÖÜÉÈÇÑØÅİıΣσςΩωЖжЯяЇї") + list("中文字符日本語テキスト한국어") + ["́", "̈", "​", "­", " ", "　", "🙂", "👍🏽"]
            + ["[UNK]", "[PAD]", "<unk>", "the", "search", "text", "and"])
    out = list(CORPUS[:8]) + ["", " ", "   ", "a", "[UNK]", "x[PAD]y [UNK]", "ΟΔΥΣΣΕΥΣ", "İ", "word" * 40, "é" * 120]
    for _ in range(n):
        out.append("".join(rng.choice(pool) for _ in range(rng.randint(1, 60))))
    # pure-ASCII lines (the native tokenizer's one-pass path for BertNormalizer -> BertPreTokenizer -> WordPiece): prose-like words,
    # every ASCII punctuation character, tabs / newlines / other control characters, upper case, words longer than 100 characters,
    # and lines that do or do not contain the first byte of an added token
    words = ["the", "quick", "Search", "TEXT", "embeddings", "cosine", "x", "fox,", "(again)", "a_b", "e-mail", "1e-5", "0xDEADBEEF", "[UNK]",
             "[PAD]", "[", "]", "<unk>", "zzzqqqxxyy", "w" * 101, "don't", "semi;colon", "q" * 100]
    ascii_pool = [chr(c) for c in range(1, 128)]
    for _ in range(n // 2):
        parts = [rng.choice(words) for _ in range(rng.randint(1, 14))]
        out.append(rng.choice([" ", "  ", "\t", "\n", " \r\n"]).join(parts))
        out.append("".join(rng.choice(ascii_pool) for _ in range(rng.randint(1, 80))))
    # Latin text below U+0300 (the one-pass BertNormalizer of the native reader): accented letters with and without canonical
    # decompositions, ligatures and digraphs, NBSP / NEL / soft hyphen / C1 controls, dotted and dotless i
    latin = [chr(c) for c in list(range(0x20, 0x7F)) + list(range(0x80, 0x250))] + ["\t", "\n", " the ", " search ", "İ", "ı", "ǅ", "ß"]
    for _ in range(n // 2):
        out.append("".join(rng.choice(latin) for _ in range(rng.randint(1, 60))))
    return out


def check(tok, path, seed):
    tok.save(str(path))
    enc = native(path)
    try:
        for text in samples(seed):
            want = tok.encode(text, add_special_tokens=False).ids
            assert enc(text) == want, repr(text)
        vs, unk, med = C.c_uint64(), C.c_int64(), C.c_uint64()
        L.check(L.lib().smt_host_tokenizer_info(enc.handle, C.byref(vs), C.byref(unk), C.byref(med)))
        vocab = tok.get_vocab()
        assert vs.value == max(vocab.values()) + 1
        lens = sorted(len(t.encode()) for t in vocab)
        assert med.value == max(1, lens[len(lens) // 2])           # model2vec-rs: median of tk.len(), bytes
    finally:
        L.lib().smt_host_tokenizer_free(enc.handle)


@pytest.mark.parametrize("lowercase,strip_accents,chinese", [(True, None, True), (False, False, True), (True, True, False),
                                                             (False, True, True)])
def test_wordpiece_bert_pipeline(tmp_path, lowercase, strip_accents, chinese):
    tok = Tokenizer(models.WordPiece(unk_token="[UNK]"))
    tok.normalizer = normalizers.BertNormalizer(lowercase=lowercase, strip_accents=strip_accents, handle_chinese_chars=chinese)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.train_from_iterator(CORPUS, trainers.WordPieceTrainer(vocab_size=400, special_tokens=["[UNK]", "[PAD]"]))
    check(tok, tmp_path / "tokenizer.json", seed=1)


def test_wordpiece_sequence_components(tmp_path):
    tok = Tokenizer(models.WordPiece(unk_token="[UNK]", max_input_chars_per_word=20))
    tok.normalizer = normalizers.Sequence([normalizers.NFD(), normalizers.StripAccents(), normalizers.Lowercase(),
                                           normalizers.Replace("ß", "ss"), normalizers.Strip()])
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.WhitespaceSplit(), pre_tokenizers.Punctuation()])
    tok.train_from_iterator(CORPUS, trainers.WordPieceTrainer(vocab_size=300, special_tokens=["[UNK]"]))
    check(tok, tmp_path / "tokenizer.json", seed=2)


def test_wordpiece_whitespace_regex_pretokenizer(tmp_path):
    tok = Tokenizer(models.WordPiece(unk_token="[UNK]"))
    tok.normalizer = normalizers.Lowercase()
    tok.pre_tokenizer = pre_tokenizers.Whitespace()          # \\w+|[^\\w\\s]+
    tok.train_from_iterator(CORPUS, trainers.WordPieceTrainer(vocab_size=300, special_tokens=["[UNK]"]))
    check(tok, tmp_path / "tokenizer.json", seed=4)


@pytest.mark.parametrize("prepend", ["always", "first", "never"])
def test_unigram_metaspace_pipeline(tmp_path, prepend):
    tok = Tokenizer(models.Unigram())
    tok.normalizer = normalizers.Sequence([normalizers.NFD(), normalizers.Lowercase()])
    tok.pre_tokenizer = pre_tokenizers.Metaspace(prepend_scheme=prepend)
    tok.train_from_iterator(CORPUS, trainers.UnigramTrainer(vocab_size=250, special_tokens=["<unk>", "[PAD]"], unk_token="<unk>"))
    check(tok, tmp_path / "tokenizer.json", seed=3)


def test_unsupported_component_fails_loudly(tmp_path):
    tok = Tokenizer(models.BPE(unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel()
    tok.train_from_iterator(CORPUS, trainers.BpeTrainer(vocab_size=300, special_tokens=["[UNK]"]))
    p = tmp_path / "tokenizer.json"
    tok.save(str(p))
    h = C.c_void_p()
    assert L.lib().smt_host_tokenizer_load(str(p).encode(), C.byref(h)) == L.SMT_E_INVALID and not h
    assert b"not supported by the native tokenizer" in L.lib().smt_last_error()
