// hf_tokenizer.cpp -- a native reader of Hugging Face `tokenizer.json` files for the host layer, so that the C++
// host (and the CLI replica) can run a real model2vec model without Python.
//
// Reference: the reference reaches its tokenizer through model2vec-rs, which loads `tokenizer.json` with the
// `tokenizers` crate and calls encode_batch_fast(.., add_special_tokens = false) (call sites src/search/mod.rs:69,
// src/cmds/search.rs:123-128,136,154).  That crate's algorithm is restated here for the component types model2vec
// models are built from -- anything else fails loudly at load time:
//   normalizers     BertNormalizer, Lowercase, NFD, StripAccents, Strip, Replace (string pattern), Sequence
//   pre_tokenizers  BertPreTokenizer, Whitespace, WhitespaceSplit, Punctuation, Metaspace, Sequence
//   models          WordPiece (greedy longest match), Unigram (Viterbi, fused unknowns)
//   added_tokens    matched on the raw text (special / normalized = false) before everything else
// Pinned, not recalled: tests/test_tokenizer.py drives this file against the `tokenizers` Python wheel -- the same
// Rust code the reference runs -- on tokenizers trained in the test and on random multilingual text.
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <unordered_map>

#include "host.h"
#include "flat_vocab.h"
#include "json.h"
#include "unicode_lower.h"

namespace semtools {
namespace {

typedef std::u32string U32;

// ------------------------------------------------------------------ Unicode helpers
template <typename R>
bool in_ranges(const R *r, size_t n, uint32_t cp)
{
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (cp > r[mid].last) lo = mid + 1;
        else if (cp < r[mid].first) hi = mid;
        else return true;
    }
    return false;
}
#define SMT_IN(table, cp) in_ranges(unicode::table, sizeof(unicode::table) / sizeof(unicode::table[0]), cp)

bool is_white_space(uint32_t c)  // Rust char::is_whitespace == the White_Space property
{
    return (c >= 0x9 && c <= 0xD) || c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) || c == 0x2028 ||
           c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}
bool is_other(uint32_t c) { return c < 0x80 ? (c < 0x20 || c == 0x7F) : SMT_IN(CAT_OTHER, c); }
bool is_punct_cat(uint32_t c)
{
    if (c < 0x80) return (c >= 0x21 && c <= 0x2F && c != '$' && c != '+') || c == ':' || c == ';' || c == '?' || c == '@' ||
                         (c >= '[' && c <= ']') || c == '_' || c == '{' || c == '}';
    return SMT_IN(CAT_PUNCT, c);
}
bool is_ascii_punct(uint32_t c) { return (c >= 0x21 && c <= 0x2F) || (c >= 0x3A && c <= 0x40) || (c >= 0x5B && c <= 0x60) || (c >= 0x7B && c <= 0x7E); }
bool is_bert_punc(uint32_t c) { return is_ascii_punct(c) || is_punct_cat(c); }
bool is_mn(uint32_t c) { return c >= 0x300 && SMT_IN(CAT_MN, c); }
bool is_regex_word(uint32_t c) { return c < 0x80 ? (isalnum((int)c) != 0 || c == '_') : SMT_IN(REGEX_WORD, c); }
bool is_chinese_char(uint32_t c)
{
    return (c >= 0x4E00 && c <= 0x9FFF) || (c >= 0x3400 && c <= 0x4DBF) || (c >= 0x20000 && c <= 0x2A6DF) || (c >= 0x2A700 && c <= 0x2B73F) ||
           (c >= 0x2B740 && c <= 0x2B81F) || (c >= 0x2B920 && c <= 0x2CEAF) || (c >= 0xF900 && c <= 0xFAFF) || (c >= 0x2F800 && c <= 0x2FA1F);
}
uint32_t ccc_of(uint32_t c)
{
    if (c < 0x300) return 0;
    const size_t n = sizeof(unicode::CCC_RUNS) / sizeof(unicode::CCC_RUNS[0]);
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (c > unicode::CCC_RUNS[mid].last) lo = mid + 1;
        else if (c < unicode::CCC_RUNS[mid].first) hi = mid;
        else return unicode::CCC_RUNS[mid].ccc;
    }
    return 0;
}

U32 decode(const std::string &s)
{
    U32 out;
    out.reserve(s.size());
    for (size_t i = 0; i < s.size();) {
        const unsigned char c = (unsigned char)s[i];
        size_t want = c < 0x80 ? 1 : (c >> 5) == 0x6 ? 2 : (c >> 4) == 0xE ? 3 : (c >> 3) == 0x1E ? 4 : 0;
        bool ok = want != 0 && i + want <= s.size();
        for (size_t k = 1; ok && k < want; ++k) ok = ((unsigned char)s[i + k] & 0xC0) == 0x80;
        if (!ok) { out.push_back(0xFFFD); ++i; continue; }
        uint32_t cp = want == 1 ? c : want == 2 ? (c & 0x1Fu) : want == 3 ? (c & 0x0Fu) : (c & 0x07u);
        for (size_t k = 1; k < want; ++k) cp = (cp << 6) | ((unsigned char)s[i + k] & 0x3Fu);
        out.push_back(cp);
        i += want;
    }
    return out;
}
void encode_cp(std::string &o, uint32_t cp)
{
    if (cp < 0x80) o.push_back((char)cp);
    else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    else { o.push_back((char)(0xF0 | (cp >> 18))); o.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
}
std::string to_utf8(const U32 &s, size_t b = 0, size_t e = (size_t)-1)
{
    std::string o;
    if (e > s.size()) e = s.size();
    for (size_t i = b; i < e; ++i) encode_cp(o, s[i]);
    return o;
}

// per-CHARACTER lower-casing (NormalizedString::lowercase maps char::to_lowercase over the chars: no Final_Sigma context)
void lower_char(uint32_t c, U32 &out)
{
    if (c < 0x80) { out.push_back(c >= 'A' && c <= 'Z' ? c + 32 : c); return; }
    if (c == 0x3A3) { out.push_back(0x3C3); return; }
    for (const unicode::LowerMulti &m : unicode::LOWER_MULTI)
        if (m.cp == c) { for (uint32_t k = 0; k < m.n; ++k) out.push_back(m.to[k]); return; }
    const size_t n = sizeof(unicode::LOWER_RUNS) / sizeof(unicode::LOWER_RUNS[0]);
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        const unicode::LowerRun &r = unicode::LOWER_RUNS[mid];
        if (c > r.last) lo = mid + 1;
        else if (c < r.first) hi = mid;
        else { out.push_back(((c - r.first) % r.stride == 0) ? (uint32_t)((int64_t)c + r.delta) : c); return; }
    }
    out.push_back(c);
}

// canonical decomposition (NFD): table + algorithmic Hangul + canonical ordering of the combining marks
void nfd(const U32 &in, U32 &out)
{
    out.clear();
    const size_t n = sizeof(unicode::NFD_TABLE) / sizeof(unicode::NFD_TABLE[0]);
    for (uint32_t c : in) {
        if (c < 0xC0) { out.push_back(c); continue; }
        if (c >= 0xAC00 && c <= 0xD7A3) {
            const uint32_t s = c - 0xAC00, l = 0x1100 + s / 588, v = 0x1161 + (s % 588) / 28, t = 0x11A7 + s % 28;
            out.push_back(l); out.push_back(v);
            if (t != 0x11A7) out.push_back(t);
            continue;
        }
        size_t lo = 0, hi = n;
        bool found = false;
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (c > unicode::NFD_TABLE[mid].cp) lo = mid + 1;
            else if (c < unicode::NFD_TABLE[mid].cp) hi = mid;
            else { for (uint32_t k = 0; k < unicode::NFD_TABLE[mid].n; ++k) out.push_back(unicode::NFD_TABLE[mid].to[k]); found = true; break; }
        }
        if (!found) out.push_back(c);
    }
    for (size_t i = 0; i < out.size();) {  // stable sort of every run of non-starters by combining class
        if (ccc_of(out[i]) == 0) { ++i; continue; }
        size_t j = i;
        while (j < out.size() && ccc_of(out[j]) != 0) ++j;
        std::stable_sort(out.begin() + i, out.begin() + j, [](uint32_t a, uint32_t b) { return ccc_of(a) < ccc_of(b); });
        i = j;
    }
}

// ------------------------------------------------------------------ components
struct Normalizer {
    enum Kind { Bert, Lower, Nfd, StripAcc, Strip, Replace, Seq } kind = Seq;
    bool clean_text = true, chinese = true, strip_accents = true, lowercase = true;  // Bert
    bool strip_left = true, strip_right = true;                                        // Strip
    U32 pattern, content;                                                              // Replace
    std::vector<Normalizer> children;
};
struct PreTokenizer {
    enum Kind { Bert, WsSplit, Whitespace, Punct, Metaspace, Seq } kind = Seq;
    uint32_t replacement = 0x2581;
    int prepend = 0;  // Metaspace: 0 always, 1 first, 2 never
    bool split = true;
    int punct_behavior = 0;  // Punctuation: 0 isolated, 1 removed, 2 merged_with_previous, 3 merged_with_next, 4 contiguous
    std::vector<PreTokenizer> children;
};

const json::Value &need(const json::Value &v, const char *k)
{
    const json::Value *x = v.get(k);
    if (!x) throw Error(std::string("tokenizer.json: missing field '") + k + "'");
    return *x;
}
bool flag(const json::Value &v, const char *k, bool dflt)
{
    const json::Value *x = v.get(k);
    return x && x->kind == json::Value::Bool ? x->b : dflt;
}

Normalizer parse_normalizer(const json::Value &v)
{
    Normalizer n;
    const std::string type = need(v, "type").s;
    if (type == "BertNormalizer") {
        n.kind = Normalizer::Bert;
        n.clean_text = flag(v, "clean_text", true);
        n.chinese = flag(v, "handle_chinese_chars", true);
        n.lowercase = flag(v, "lowercase", true);
        const json::Value *sa = v.get("strip_accents");
        n.strip_accents = sa && sa->kind == json::Value::Bool ? sa->b : n.lowercase;  // None -> follows lowercase
    } else if (type == "Lowercase") n.kind = Normalizer::Lower;
    else if (type == "NFD") n.kind = Normalizer::Nfd;
    else if (type == "StripAccents") n.kind = Normalizer::StripAcc;
    else if (type == "Strip") {
        n.kind = Normalizer::Strip;
        n.strip_left = flag(v, "strip_left", true);
        n.strip_right = flag(v, "strip_right", true);
    } else if (type == "Replace") {
        n.kind = Normalizer::Replace;
        const json::Value &p = need(v, "pattern");
        const json::Value *str = p.get("String");
        if (!str) throw Error("tokenizer.json: Replace normalizer with a Regex pattern is not supported by the native tokenizer");
        n.pattern = decode(str->s);
        n.content = decode(need(v, "content").s);
    } else if (type == "Sequence") {
        n.kind = Normalizer::Seq;
        for (auto &c : need(v, "normalizers").arr) n.children.push_back(parse_normalizer(c));
    } else {
        throw Error("tokenizer.json: normalizer type '" + type + "' is not supported by the native tokenizer "
                    "(supported: BertNormalizer, Lowercase, NFD, StripAccents, Strip, Replace, Sequence)");
    }
    return n;
}

PreTokenizer parse_pre(const json::Value &v)
{
    PreTokenizer p;
    const std::string type = need(v, "type").s;
    if (type == "BertPreTokenizer") p.kind = PreTokenizer::Bert;
    else if (type == "WhitespaceSplit") p.kind = PreTokenizer::WsSplit;
    else if (type == "Whitespace") p.kind = PreTokenizer::Whitespace;
    else if (type == "Punctuation") {
        p.kind = PreTokenizer::Punct;
        const json::Value *b = v.get("behavior");
        const std::string bs = b ? b->s : "Isolated";
        p.punct_behavior = bs == "Isolated" ? 0 : bs == "Removed" ? 1 : bs == "MergedWithPrevious" ? 2 : bs == "MergedWithNext" ? 3 : 4;
    } else if (type == "Metaspace") {
        p.kind = PreTokenizer::Metaspace;
        if (const json::Value *r = v.get("replacement")) { const U32 u = decode(r->s); if (!u.empty()) p.replacement = u[0]; }
        const json::Value *ps = v.get("prepend_scheme");
        if (ps) p.prepend = ps->s == "always" ? 0 : ps->s == "first" ? 1 : 2;
        else p.prepend = flag(v, "add_prefix_space", true) ? 0 : 2;   // older files
        p.split = flag(v, "split", true);
    } else if (type == "Sequence") {
        p.kind = PreTokenizer::Seq;
        for (auto &c : need(v, "pretokenizers").arr) p.children.push_back(parse_pre(c));
    } else {
        throw Error("tokenizer.json: pre_tokenizer type '" + type + "' is not supported by the native tokenizer "
                    "(supported: BertPreTokenizer, Whitespace, WhitespaceSplit, Punctuation, Metaspace, Sequence)");
    }
    return p;
}

// BertNormalizer in ONE pass for a text whose code points are all below U+0300 (ASCII, Latin-1, Latin Extended, IPA: no combining
// mark, no CJK): per character exactly what the four passes of apply_normalizer's Bert case do one after the other -- clean_text
// drops NUL / controls / format characters and turns white space into ' '; Chinese characters cannot occur; strip_accents keeps the
// base of the canonical decomposition (every mark such a character decomposes into is a non-spacing mark U+0300..U+036F, so the
// canonical reordering only permutes characters that are removed); lower-casing is per character.
bool bert_normalize_below_0300(const Normalizer &n, const U32 &in, U32 &out)
{
    for (uint32_t c : in) if (c >= 0x300) return false;
    out.clear();
    out.reserve(in.size());
    const size_t n_nfd = sizeof(unicode::NFD_TABLE) / sizeof(unicode::NFD_TABLE[0]);
    for (uint32_t c : in) {
        if (n.clean_text) {
            const bool tnr = c == '\t' || c == '\n' || c == '\r';
            if (c == 0 || (!tnr && is_other(c))) continue;
            if (tnr || is_white_space(c)) c = ' ';
        }
        if (c < 0x80) { out.push_back(n.lowercase && c >= 'A' && c <= 'Z' ? c + 32 : c); continue; }
        if (n.strip_accents && c >= 0xC0) {
            size_t lo = 0, hi = n_nfd;
            bool found = false;
            while (lo < hi) {
                const size_t mid = (lo + hi) / 2;
                if (c > unicode::NFD_TABLE[mid].cp) lo = mid + 1;
                else if (c < unicode::NFD_TABLE[mid].cp) hi = mid;
                else {
                    for (uint32_t k = 0; k < unicode::NFD_TABLE[mid].n; ++k) {
                        const uint32_t d = unicode::NFD_TABLE[mid].to[k];
                        if (is_mn(d)) continue;
                        if (n.lowercase) lower_char(d, out); else out.push_back(d);
                    }
                    found = true;
                    break;
                }
            }
            if (found) continue;
        }
        if (n.lowercase) lower_char(c, out); else out.push_back(c);
    }
    return true;
}

void apply_normalizer(const Normalizer &n, U32 &s)
{
    U32 t;
    switch (n.kind) {
        case Normalizer::Seq:
            for (auto &c : n.children) apply_normalizer(c, s);
            return;
        case Normalizer::Lower:
            for (uint32_t c : s) lower_char(c, t);
            s.swap(t);
            return;
        case Normalizer::Nfd:
            nfd(s, t);
            s.swap(t);
            return;
        case Normalizer::StripAcc:
            for (uint32_t c : s) if (!is_mn(c)) t.push_back(c);
            s.swap(t);
            return;
        case Normalizer::Strip: {
            size_t b = 0, e = s.size();
            if (n.strip_left) while (b < e && is_white_space(s[b])) ++b;
            if (n.strip_right) while (e > b && is_white_space(s[e - 1])) --e;
            s = s.substr(b, e - b);
            return;
        }
        case Normalizer::Replace: {
            if (n.pattern.empty()) return;
            for (size_t i = 0; i < s.size();) {
                if (s.compare(i, n.pattern.size(), n.pattern) == 0) { t += n.content; i += n.pattern.size(); }
                else t.push_back(s[i++]);
            }
            s.swap(t);
            return;
        }
        case Normalizer::Bert: {
            if (bert_normalize_below_0300(n, s, t)) { s.swap(t); return; }
            if (n.clean_text) {  // drop NUL / U+FFFD / control characters, every white space becomes ' '
                for (uint32_t c : s) {
                    const bool ws = c == '\t' || c == '\n' || c == '\r' || is_white_space(c);
                    const bool control = !(c == '\t' || c == '\n' || c == '\r') && is_other(c);
                    if (c == 0 || c == 0xFFFD || control) continue;
                    t.push_back(ws ? ' ' : c);
                }
                s.swap(t);
                t.clear();
            }
            if (n.chinese) {
                for (uint32_t c : s) {
                    if (is_chinese_char(c)) { t.push_back(' '); t.push_back(c); t.push_back(' '); }
                    else t.push_back(c);
                }
                s.swap(t);
                t.clear();
            }
            if (n.strip_accents) {
                U32 d;
                nfd(s, d);
                for (uint32_t c : d) if (!is_mn(c)) t.push_back(c);
                s.swap(t);
                t.clear();
            }
            if (n.lowercase) {
                for (uint32_t c : s) lower_char(c, t);
                s.swap(t);
            }
            return;
        }
    }
}

// pieces are [begin, end) spans of one buffer; a pre-tokenizer maps every span to sub-spans (Metaspace rewrites the text,
// so pieces are plain strings here)
void apply_pre(const PreTokenizer &p, std::vector<U32> &pieces, bool first_section)
{
    std::vector<U32> out;
    switch (p.kind) {
        case PreTokenizer::Seq:
            for (auto &c : p.children) apply_pre(c, pieces, first_section);
            return;
        case PreTokenizer::WsSplit:
            for (auto &w : pieces) {
                size_t i = 0;
                while (i < w.size()) {
                    while (i < w.size() && is_white_space(w[i])) ++i;
                    const size_t b = i;
                    while (i < w.size() && !is_white_space(w[i])) ++i;
                    if (i > b) out.push_back(w.substr(b, i - b));
                }
            }
            break;
        case PreTokenizer::Whitespace:  // the regex \w+|[^\w\s]+ : runs of word characters, runs of everything else but white space
            for (auto &w : pieces) {
                size_t i = 0;
                while (i < w.size()) {
                    if (is_white_space(w[i])) { ++i; continue; }
                    const bool word = is_regex_word(w[i]);
                    const size_t b = i;
                    while (i < w.size() && !is_white_space(w[i]) && is_regex_word(w[i]) == word) ++i;
                    out.push_back(w.substr(b, i - b));
                }
            }
            break;
        case PreTokenizer::Bert:
            for (auto &w : pieces) {
                size_t i = 0;
                while (i < w.size()) {
                    if (is_white_space(w[i])) { ++i; continue; }
                    if (is_bert_punc(w[i])) { out.push_back(U32(1, w[i])); ++i; continue; }
                    const size_t b = i;
                    while (i < w.size() && !is_white_space(w[i]) && !is_bert_punc(w[i])) ++i;
                    out.push_back(w.substr(b, i - b));
                }
            }
            break;
        case PreTokenizer::Punct:
            for (auto &w : pieces) {
                // split on is_punctuation with the configured behaviour (Isolated is what every shipped tokenizer uses)
                std::vector<std::pair<U32, bool>> parts;  // (text, is_delimiter)
                size_t i = 0;
                while (i < w.size()) {
                    if (is_bert_punc(w[i])) { parts.emplace_back(U32(1, w[i]), true); ++i; continue; }
                    const size_t b = i;
                    while (i < w.size() && !is_bert_punc(w[i])) ++i;
                    parts.emplace_back(w.substr(b, i - b), false);
                }
                if (p.punct_behavior == 0) for (auto &pt : parts) out.push_back(pt.first);
                else if (p.punct_behavior == 1) { for (auto &pt : parts) if (!pt.second) out.push_back(pt.first); }
                else if (p.punct_behavior == 2) {  // merged with previous
                    for (auto &pt : parts) { if (pt.second && !out.empty()) out.back() += pt.first; else out.push_back(pt.first); }
                } else if (p.punct_behavior == 3) {  // merged with next
                    U32 carry;
                    for (auto &pt : parts) { if (pt.second) carry += pt.first; else { out.push_back(carry + pt.first); carry.clear(); } }
                    if (!carry.empty()) out.push_back(carry);
                } else {  // contiguous: runs of delimiters stay together
                    for (auto &pt : parts) {
                        if (pt.second && !out.empty() && !out.back().empty() && is_bert_punc(out.back()[0])) out.back() += pt.first;
                        else out.push_back(pt.first);
                    }
                }
            }
            break;
        case PreTokenizer::Metaspace:
            for (size_t wi = 0; wi < pieces.size(); ++wi) {
                U32 w;
                for (uint32_t c : pieces[wi]) w.push_back(c == ' ' ? p.replacement : c);
                const bool may_prepend = p.prepend == 0 || (p.prepend == 1 && first_section && wi == 0);
                if (may_prepend && (w.empty() || w[0] != p.replacement)) w.insert(w.begin(), p.replacement);
                if (!p.split) { out.push_back(w); continue; }
                size_t b = 0;  // MergedWithNext on the replacement: every replacement starts a new piece
                for (size_t i = 1; i <= w.size(); ++i)
                    if (i == w.size() || w[i] == p.replacement) { if (i > b) out.push_back(w.substr(b, i - b)); b = i; }
            }
            break;
    }
    pieces.swap(out);
}

struct AddedToken {
    U32 content;
    std::string bytes;   // the content as bytes when it is pure ASCII and matched on the raw text (encode_ascii)
    uint32_t id = 0;
    bool single_word = false, lstrip = false, rstrip = false, normalized = false;
};

}  // namespace

// ------------------------------------------------------------------ the tokenizer
class HfTokenizer : public Tokenizer {
public:
    explicit HfTokenizer(const std::string &path)
    {
        // The vocabulary (500 k entries in potion-multilingual) stays out of the DOM -- 150 B per entry as json::Values,
        // most of the load time -- and is read straight into the lookup table while the parser passes over it.
        const std::string text = read_to_string(path);
        std::vector<size_t> lens;
        bool vocab_is_array = false, vocab_seen = false;
        double min_score = 0.0;
        json::Parser parser(text);
        parser.read_member("vocab", 2, [&](json::Parser &p) {     // root object -> "model" -> "vocab"
            vocab_seen = true;
            vocab_.reserve_bytes(text.size() / 2);
            lens.reserve(text.size() / 24);
            if (p.accept('{')) {                                   // WordPiece (and BPE, refused below): {"piece": id, ...}
                if (p.accept('}')) return;
                for (;;) {
                    const std::string piece = p.string();
                    p.expect(':');
                    const uint32_t id = (uint32_t)p.number_f64();
                    size_ = std::max<uint64_t>(size_, (uint64_t)id + 1);
                    lens.push_back(piece.size());
                    vocab_.add(piece, id);
                    if (p.accept(',')) continue;
                    p.expect('}');
                    return;
                }
            }
            vocab_is_array = true;                                 // Unigram: [["piece", score], ...]
            p.expect('[');
            if (p.accept(']')) return;
            uint32_t id = 0;
            for (;;) {
                p.expect('[');
                const std::string piece = p.string();
                p.expect(',');
                const double sc = p.number_f64();
                if (!p.accept(']')) throw Error("tokenizer.json: malformed Unigram vocab entry");
                scores_.push_back(sc);
                max_piece_bytes_ = std::max(max_piece_bytes_, piece.size());
                lens.push_back(piece.size());
                if (id == 0 || sc < min_score) min_score = sc;
                vocab_.add(piece, id);                             // (a repeated piece: the later id wins, like the crate's HashMap insert)
                ++id;
                if (p.accept(',')) continue;
                p.expect(']');
                return;
            }
        });
        const json::Value root = parser.parse();
        if (!vocab_seen) throw Error("tokenizer.json: model.vocab is missing");
        vocab_.build();
        if (const json::Value *n = root.get("normalizer"); n && n->kind == json::Value::Object) { norm_ = parse_normalizer(*n); has_norm_ = true; }
        if (const json::Value *p = root.get("pre_tokenizer"); p && p->kind == json::Value::Object) { pre_ = parse_pre(*p); has_pre_ = true; }
        const json::Value &model = need(root, "model");
        const json::Value *type = model.get("type");
        const std::string kind = type ? type->s : (vocab_is_array ? "Unigram" : "WordPiece");
        if (kind == "WordPiece") {
            if (vocab_is_array) throw Error("tokenizer.json: a WordPiece vocabulary must be an object");
            wordpiece_ = true;
            prefix_ = model.get("continuing_subword_prefix") ? model.get("continuing_subword_prefix")->s : "##";
            max_chars_ = model.get("max_input_chars_per_word") ? (size_t)model.get("max_input_chars_per_word")->as_u64() : 100;
            const std::string unk = model.get("unk_token") ? model.get("unk_token")->s : "[UNK]";
            const int64_t unk_id = vocab_.find(unk);
            if (unk_id < 0) throw Error("tokenizer.json: WordPiece unk_token '" + unk + "' is not in the vocabulary");
            unk_ = (uint32_t)unk_id;
        } else if (kind == "Unigram") {
            if (!vocab_is_array) throw Error("tokenizer.json: a Unigram vocabulary must be an array of [piece, score]");
            if (flag(model, "byte_fallback", false)) throw Error("tokenizer.json: Unigram byte_fallback is not supported by the native tokenizer");
            size_ = scores_.size();
            if (const json::Value *u = model.get("unk_id"); u && u->kind != json::Value::Null) unk_ = (uint32_t)u->as_u64();
            unk_score_ = min_score - 10.0;  // K_UNK_PENALTY
        } else {
            throw Error("tokenizer.json: model type '" + kind + "' is not supported by the native tokenizer (WordPiece, Unigram)");
        }
        if (!lens.empty()) {  // model2vec-rs: median of tk.len() over the vocabulary (bytes): the upper median of the sorted lengths
            std::nth_element(lens.begin(), lens.begin() + lens.size() / 2, lens.end());
            median_ = std::max<size_t>(1, lens[lens.size() / 2]);
        }
        if (const json::Value *added = root.get("added_tokens"))
            for (auto &a : added->arr) {
                AddedToken t;
                t.content = decode(need(a, "content").s);
                t.id = (uint32_t)need(a, "id").as_u64();
                t.single_word = flag(a, "single_word", false);
                t.lstrip = flag(a, "lstrip", false);
                t.rstrip = flag(a, "rstrip", false);
                t.normalized = flag(a, "normalized", false);
                size_ = std::max<uint64_t>(size_, (uint64_t)t.id + 1);
                if (!t.content.empty()) added_.push_back(std::move(t));
            }
        for (AddedToken &t : added_) {
            if (t.normalized) continue;
            bool ascii = true;
            for (uint32_t c : t.content) ascii = ascii && c < 0x80;
            if (!ascii) continue;   // (cannot stand in a pure-ASCII line)
            t.bytes.assign(t.content.begin(), t.content.end());
            added_first_[t.content[0]] = true;
            // lstrip / rstrip tokens swallow neighbouring white space, which cannot change what a line WITHOUT the token encodes to;
            // single_word tokens that do not match because of their neighbours still send the line to the general path (rare, correct)
        }
        for (const AddedToken &t : added_) if (!t.bytes.empty()) added_ascii_.push_back(t);
        ascii_fast_ = wordpiece_ && has_pre_ && pre_.kind == PreTokenizer::Bert && (!has_norm_ || norm_.kind == Normalizer::Bert);
    }

    void encode(const std::string &text, std::vector<uint32_t> &ids) const override
    {
        if (ascii_fast_ && encode_ascii(text, ids)) return;
        const U32 raw = decode(text);
        // ---- added tokens are cut out of the RAW text first (leftmost, longest at a position); the spans between
        // them go through normalizer -> pre-tokenizer -> model
        size_t pos = 0, section_begin = 0;
        bool first_section = true;
        auto flush = [&](size_t b, size_t e) {
            if (e > b) encode_section(raw.substr(b, e - b), first_section, ids);
            if (e > b) first_section = false;
        };
        while (pos < raw.size() && !added_.empty()) {
            const AddedToken *hit = nullptr;
            for (const AddedToken &t : added_) {
                if (t.normalized) continue;  // (normalized added tokens are matched on the normalised text: rare, see encode_section)
                if (raw.compare(pos, t.content.size(), t.content) != 0) continue;
                if (t.single_word) {
                    const bool lb = pos == 0 || !is_word_char(raw[pos - 1]);
                    const bool rb = pos + t.content.size() >= raw.size() || !is_word_char(raw[pos + t.content.size()]);
                    if (!lb || !rb) continue;
                }
                if (!hit || t.content.size() > hit->content.size()) hit = &t;
            }
            if (!hit) { ++pos; continue; }
            size_t b = pos, e = pos + hit->content.size();
            if (hit->lstrip) while (b > section_begin && is_white_space(raw[b - 1])) --b;
            if (hit->rstrip) while (e < raw.size() && is_white_space(raw[e])) ++e;
            flush(section_begin, b);
            ids.push_back(hit->id);
            first_section = false;
            pos = section_begin = e;
        }
        flush(section_begin, raw.size());
    }

    std::optional<uint32_t> unk_id() const override { return unk_; }
    size_t median_token_length() const override { return median_; }
    uint64_t vocab_size() const override { return size_; }
    size_t lines_per_thread() const override { return 1024; }

private:
    static bool is_word_char(uint32_t c) { return c == '_' || (c < 0x80 ? isalnum((int)c) != 0 : !is_white_space(c) && !is_punct_cat(c)); }

    void encode_section(U32 s, bool first_section, std::vector<uint32_t> &ids) const
    {
        if (has_norm_) apply_normalizer(norm_, s);
        if (wordpiece_ && has_pre_ && pre_.kind == PreTokenizer::Bert) { wordpiece_bert_words(s, ids); return; }
        std::vector<U32> pieces(1, std::move(s));
        if (has_pre_) apply_pre(pre_, pieces, first_section);
        for (const U32 &w : pieces) {
            if (w.empty()) continue;
            if (wordpiece_) wordpiece(w, ids);
            else unigram(w, ids);
        }
    }

    // ---- the common case in one pass over the BYTES: a pure-ASCII line through BertNormalizer -> BertPreTokenizer -> WordPiece (the
    // pipeline of the model2vec "potion" English models).  Same steps as the general path below restricted to code points < 0x80 --
    // clean_text drops NUL / controls and turns \t \n \r into ' ', accents and Chinese characters cannot occur, lower-casing is A-Z,
    // the pre-tokenizer isolates ASCII punctuation and splits at white space, WordPiece matches greedily -- without the u32 copy of
    // the text, a string per piece and three vectors per word (3.9 -> ~0.8 us per line of prose; tests/test_tokenizer.py runs both
    // paths against the `tokenizers` wheel).  Returns false -- nothing appended -- when the line is not its case: a byte >= 0x80, or a
    // place where an added token stands (those are matched on the raw text first: the general path does it).
    bool encode_ascii(const std::string &text, std::vector<uint32_t> &ids) const
    {
        static thread_local std::string norm;
        norm.clear();
        const bool bert_norm = has_norm_;
        for (size_t at = 0; at < text.size(); ++at) {
            const unsigned char c = (unsigned char)text[at];
            if (c >= 0x80) return false;
            if (added_first_[c]) {   // ('[' is in every Markdown or source line: only a token that really stands here sends the line away)
                for (const AddedToken &t : added_ascii_)
                    if (text.compare(at, t.bytes.size(), t.bytes) == 0) return false;
            }
            if (!bert_norm) { norm.push_back((char)c); continue; }
            if (norm_.clean_text) {
                if (c == '\t' || c == '\n' || c == '\r') { norm.push_back(' '); continue; }
                if (c < 0x20 || c == 0x7F) continue;   // NUL and the controls
            }
            norm.push_back(norm_.lowercase && c >= 'A' && c <= 'Z' ? (char)(c + 32) : (char)c);
        }
        static thread_local std::string cand;
        const char *p = norm.data();
        const size_t n = norm.size();
        auto is_ws = [](unsigned char c) { return (c >= 0x9 && c <= 0xD) || c == 0x20; };
        auto is_punc = [](unsigned char c) { return (c >= 0x21 && c <= 0x2F) || (c >= 0x3A && c <= 0x40) || (c >= 0x5B && c <= 0x60) || (c >= 0x7B && c <= 0x7E); };
        size_t i = 0;
        while (i < n) {
            if (is_ws((unsigned char)p[i])) { ++i; continue; }
            const size_t b = i;
            if (is_punc((unsigned char)p[i])) ++i;
            else while (i < n && !is_ws((unsigned char)p[i]) && !is_punc((unsigned char)p[i])) ++i;
            // WordPiece::tokenize on [b, i): one byte per character here
            const size_t len = i - b, mark = ids.size();
            if (len > max_chars_) { ids.push_back(*unk_); continue; }
            size_t start = 0;
            bool bad = false;
            while (start < len) {
                size_t end = len;
                int64_t hit = -1;
                if (start == 0) {
                    for (; end > start; --end)
                        if ((hit = vocab_.find(p + b, end)) >= 0) break;
                } else {
                    cand.assign(prefix_);
                    cand.append(p + b + start, len - start);
                    for (; end > start; --end)
                        if ((hit = vocab_.find(cand.data(), prefix_.size() + (end - start))) >= 0) break;
                }
                if (hit < 0) { bad = true; break; }
                ids.push_back((uint32_t)hit);
                start = end;
            }
            if (bad) { ids.resize(mark); ids.push_back(*unk_); }
        }
        return true;
    }

    // BertPreTokenizer + WordPiece over the normalised text without a string per word: the words are spans of `s` (white space
    // skipped, every punctuation character a word of its own: apply_pre's Bert case), each converted once into a per-thread byte
    // buffer with the byte offset of every character, then matched greedily like wordpiece() below.  (Lines with one accented
    // letter take this path -- most lines of most European languages: 3.9 -> 2 us per line.)
    void wordpiece_bert_words(const U32 &s, std::vector<uint32_t> &ids) const
    {
        static thread_local std::string bytes, cand;
        static thread_local std::vector<uint32_t> off;
        size_t i = 0;
        const size_t n = s.size();
        while (i < n) {
            if (is_white_space(s[i])) { ++i; continue; }
            const size_t b = i;
            if (is_bert_punc(s[i])) ++i;
            else while (i < n && !is_white_space(s[i]) && !is_bert_punc(s[i])) ++i;
            const size_t len = i - b, mark = ids.size();
            if (len > max_chars_) { ids.push_back(*unk_); continue; }
            bytes.clear();
            off.resize(len + 1);
            for (size_t c = 0; c < len; ++c) { off[c] = (uint32_t)bytes.size(); encode_cp(bytes, s[b + c]); }
            off[len] = (uint32_t)bytes.size();
            size_t start = 0;
            bool bad = false;
            while (start < len) {
                size_t end = len;
                int64_t hit = -1;
                if (start == 0) {
                    for (; end > start; --end)
                        if ((hit = vocab_.find(bytes.data(), off[end])) >= 0) break;
                } else {
                    cand.assign(prefix_);
                    cand.append(bytes, off[start], std::string::npos);
                    for (; end > start; --end)
                        if ((hit = vocab_.find(cand.data(), prefix_.size() + (off[end] - off[start]))) >= 0) break;
                }
                if (hit < 0) { bad = true; break; }
                ids.push_back((uint32_t)hit);
                start = end;
            }
            if (bad) { ids.resize(mark); ids.push_back(*unk_); }
        }
    }

    // WordPiece::tokenize: greedy longest match; a word with an unmatched tail (or too many chars) is ONE unk token
    void wordpiece(const U32 &w, std::vector<uint32_t> &ids) const
    {
        if (w.size() > max_chars_) { ids.push_back(*unk_); return; }
        std::vector<size_t> off(w.size() + 1, 0);  // byte offset of every char
        const std::string bytes = to_utf8(w);
        { size_t b = 0; for (size_t i = 0; i < w.size(); ++i) { off[i] = b; b += w[i] < 0x80 ? 1 : w[i] < 0x800 ? 2 : w[i] < 0x10000 ? 3 : 4; } off[w.size()] = b; }
        std::vector<uint32_t> sub;
        size_t start = 0;
        std::string cand;
        while (start < w.size()) {
            size_t end = w.size();
            bool found = false;
            while (start < end) {
                cand.clear();
                if (start > 0) cand = prefix_;
                cand.append(bytes, off[start], off[end] - off[start]);
                const int64_t hit = vocab_.find(cand);
                if (hit >= 0) { sub.push_back((uint32_t)hit); found = true; break; }
                --end;
            }
            if (!found) { ids.push_back(*unk_); return; }
            start = end;
        }
        ids.insert(ids.end(), sub.begin(), sub.end());
    }

    // Unigram::encode_optimized: Viterbi over the byte positions, unknown characters cost min_score - 10 and
    // consecutive unknowns are fused into one unk id (fuse_unk = true)
    void unigram(const U32 &w, std::vector<uint32_t> &ids) const
    {
        const std::string s = to_utf8(w);
        const size_t n = s.size();
        struct Node { double score = 0.0; size_t from = (size_t)-1; uint32_t id = 0; bool unk = false; bool set = false; };
        std::vector<Node> best(n + 1);
        best[0].set = true;
        size_t at = 0;
        while (at < n) {
            const unsigned char c0 = (unsigned char)s[at];
            const size_t mblen = c0 < 0x80 ? 1 : (c0 >> 5) == 0x6 ? 2 : (c0 >> 4) == 0xE ? 3 : 4;
            const double here = best[at].score;
            bool single = false;
            for (size_t len = 1; at + len <= n && len <= max_piece_bytes_; ++len) {
                if (at + len < n && ((unsigned char)s[at + len] & 0xC0) == 0x80) continue;  // not a character boundary
                const int64_t hit = vocab_.find(s.data() + at, len);
                if (hit < 0) continue;
                Node &t = best[at + len];
                const double cand = scores_[(size_t)hit] + here;
                if (!t.set || cand > t.score) { t.score = cand; t.from = at; t.id = (uint32_t)hit; t.unk = false; t.set = true; }
                if (len == mblen) single = true;
            }
            if (!single) {
                Node &t = best[at + mblen];
                const double cand = unk_score_ + here;
                if (!t.set || cand > t.score) { t.score = cand; t.from = at; t.id = unk_ ? *unk_ : 0; t.unk = true; t.set = true; }
            }
            at += mblen;
        }
        std::vector<std::pair<uint32_t, bool>> rev;
        for (size_t e = n; e > 0;) { const Node &t = best[e]; rev.emplace_back(t.id, t.unk); e = t.from; }
        bool prev_unk = false;
        for (size_t i = rev.size(); i-- > 0;) {
            if (rev[i].second && prev_unk) continue;  // fused
            if (rev[i].second && !unk_) throw Error("Unigram tokenizer met an unknown character but the model has no unk_id");
            ids.push_back(rev[i].first);
            prev_unk = rev[i].second;
        }
    }

    Normalizer norm_;
    PreTokenizer pre_;
    bool has_norm_ = false, has_pre_ = false, wordpiece_ = false;
    bool ascii_fast_ = false;          // BertNormalizer (or none) -> BertPreTokenizer -> WordPiece: encode_ascii applies
    bool added_first_[128] = {};       // ASCII bytes a (non-normalized, pure-ASCII) added token starts with
    std::vector<AddedToken> added_ascii_;   // those tokens
    FlatVocab vocab_;
    std::vector<double> scores_;
    std::string prefix_ = "##";
    size_t max_chars_ = 100, max_piece_bytes_ = 0, median_ = 5;
    std::optional<uint32_t> unk_;
    double unk_score_ = 0.0;
    uint64_t size_ = 0;
    std::vector<AddedToken> added_;
};

std::unique_ptr<Tokenizer> make_hf_tokenizer(const std::string &tokenizer_json_path)
{
    return std::make_unique<HfTokenizer>(tokenizer_json_path);
}

}  // namespace semtools
