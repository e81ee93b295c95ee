// host.cpp -- C++ mirror of the reference's search module and workspace store on
// top of the C ABI (see host.h).  String handling, file I/O and bookkeeping
// only: every floating-point result comes from libsemtools_hip's kernels.
#include "host.h"
#include "unicode_lower.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <exception>
#include <fstream>
#include <sstream>
#include <thread>

#include "flat_vocab.h"
#include "fmt.h"
#include "json.h"

namespace semtools {

static void check(int rc, const char *what)
{
    if (rc != SMT_OK) throw Error(std::string(what) + ": " + smt_last_error());
}

// ------------------------------------------------------------------ strings / files
std::vector<std::string> lines_of(const std::string &content)
{
    std::vector<std::string> out;
    size_t start = 0;
    const size_t n = content.size();
    while (start < n) {
        size_t nl = content.find('\n', start);
        if (nl == std::string::npos) {
            out.emplace_back(content, start, n - start);  // last line, no terminator: kept verbatim
            break;
        }
        size_t end = nl;
        if (end > start && content[end - 1] == '\r') --end;  // "\r\n" is one line ending
        out.emplace_back(content, start, end - start);
        start = nl + 1;
    }
    return out;
}

static size_t utf8_len(unsigned char c) { return c < 0x80 ? 1 : (c >> 5) == 0x6 ? 2 : (c >> 4) == 0xE ? 3 : (c >> 3) == 0x1E ? 4 : 1; }

// ---- Unicode lower-casing with Rust's str::to_lowercase semantics (reference src/search/mod.rs:63): the full
// lower-case mapping of every code point (incl. the one multi-code-point case, U+0130 -> "i" + U+0307) and the
// Final_Sigma rule (U+03A3 -> U+03C2 at the end of a word, U+03C3 otherwise).  Table driven (unicode_lower.h,
// generated from the UCD): the process locale is never touched -- a library has no business calling setlocale.
static bool in_ranges(const unicode::CpRange *r, size_t n, uint32_t cp)
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
static bool is_cased(uint32_t cp)
{
    if (cp < 0x80) return (cp >= 'A' && cp <= 'Z') || (cp >= 'a' && cp <= 'z');
    return in_ranges(unicode::CASED, sizeof(unicode::CASED) / sizeof(unicode::CASED[0]), cp);
}
static bool is_case_ignorable(uint32_t cp)
{
    return in_ranges(unicode::CASE_IGNORABLE, sizeof(unicode::CASE_IGNORABLE) / sizeof(unicode::CASE_IGNORABLE[0]), cp);
}
static uint32_t lower_simple(uint32_t cp)
{
    const size_t n = sizeof(unicode::LOWER_RUNS) / sizeof(unicode::LOWER_RUNS[0]);
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        const unicode::LowerRun &r = unicode::LOWER_RUNS[mid];
        if (cp > r.last) lo = mid + 1;
        else if (cp < r.first) hi = mid;
        else return ((cp - r.first) % r.stride == 0) ? (uint32_t)((int64_t)cp + r.delta) : cp;
    }
    return cp;
}
static void push_utf8(std::string &out, uint32_t cp)
{
    if (cp < 0x80) out.push_back((char)cp);
    else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
    else { out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
}
// decode the code point starting at s[i]; *len = its byte length (malformed bytes pass through one by one)
static uint32_t decode_utf8(const std::string &s, size_t i, size_t *len)
{
    const unsigned char c = (unsigned char)s[i];
    const size_t want = utf8_len(c);
    if (c < 0x80 || want == 1 || i + want > s.size()) { *len = 1; return c < 0x80 ? c : 0xFFFFFFFFu; }
    for (size_t k = 1; k < want; ++k)
        if (((unsigned char)s[i + k] & 0xC0) != 0x80) { *len = 1; return 0xFFFFFFFFu; }
    *len = want;
    if (want == 2) return ((c & 0x1Fu) << 6) | ((unsigned char)s[i + 1] & 0x3Fu);
    if (want == 3) return ((c & 0x0Fu) << 12) | (((unsigned char)s[i + 1] & 0x3Fu) << 6) | ((unsigned char)s[i + 2] & 0x3Fu);
    return ((c & 0x07u) << 18) | (((unsigned char)s[i + 1] & 0x3Fu) << 12) | (((unsigned char)s[i + 2] & 0x3Fu) << 6) | ((unsigned char)s[i + 3] & 0x3Fu);
}

std::string to_lowercase(const std::string &s)
{
    std::string out;
    out.reserve(s.size());
    for (size_t i = 0; i < s.size();) {
        const unsigned char c = (unsigned char)s[i];
        if (c < 0x80) { out.push_back((char)((c >= 'A' && c <= 'Z') ? c + 32 : c)); ++i; continue; }
        size_t len = 1;
        const uint32_t cp = decode_utf8(s, i, &len);
        if (cp == 0xFFFFFFFFu) { out.push_back((char)c); ++i; continue; }  // malformed: copied verbatim
        if (cp == 0x03A3) {
            // Final_Sigma: preceded by a cased letter (skipping case-ignorables) and NOT followed by one
            bool before = false, after = false;
            for (size_t k = i; k > 0;) {
                size_t b = k - 1;
                while (b > 0 && ((unsigned char)s[b] & 0xC0) == 0x80) --b;
                size_t l2 = 1;
                const uint32_t p = decode_utf8(s, b, &l2);
                k = b;
                if (p != 0xFFFFFFFFu && is_case_ignorable(p)) continue;
                before = p != 0xFFFFFFFFu && is_cased(p);
                break;
            }
            for (size_t k = i + len; k < s.size();) {
                size_t l2 = 1;
                const uint32_t n = decode_utf8(s, k, &l2);
                k += l2;
                if (n != 0xFFFFFFFFu && is_case_ignorable(n)) continue;
                after = n != 0xFFFFFFFFu && is_cased(n);
                break;
            }
            push_utf8(out, (before && !after) ? 0x03C2u : 0x03C3u);
            i += len;
            continue;
        }
        bool multi = false;
        for (const unicode::LowerMulti &m : unicode::LOWER_MULTI)
            if (m.cp == cp) {
                for (uint32_t k = 0; k < m.n; ++k) push_utf8(out, m.to[k]);
                multi = true;
                break;
            }
        if (!multi) {
            const uint32_t lo = lower_simple(cp);
            if (lo == cp) out.append(s, i, len);
            else push_utf8(out, lo);
        }
        i += len;
    }
    return out;
}

std::string read_to_string(const std::string &path)
{
    // one read into a string sized from fstat (an ostringstream << rdbuf() copies a 30 MB tokenizer.json twice)
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error(std::string(strerror(errno)) + " (os error " + std::to_string(errno) + "): " + path);
    struct stat st;
    std::string out;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) out.resize((size_t)st.st_size);
    size_t got = 0;
    for (;;) {
        if (got == out.size()) out.resize(out.size() < 4096 ? 65536 : out.size() * 2);   // unknown size (pipe) or a file that grew
        const ssize_t n = read(fd, &out[got], out.size() - got);
        if (n < 0) {
            if (errno == EINTR) continue;
            const int e = errno;
            close(fd);
            throw Error(std::string(strerror(e)) + " (os error " + std::to_string(e) + "): " + path);
        }
        if (n == 0) break;
        got += (size_t)n;
    }
    close(fd);
    out.resize(got);
    return out;
}

static void write_file_atomic(const std::string &path, const std::string &data)
{
    const std::string tmp = path + ".tmp";
    {
        std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
        if (!f) throw Error("cannot write " + tmp + ": " + strerror(errno));
        f.write(data.data(), (std::streamsize)data.size());
        if (!f) throw Error("short write to " + tmp);
    }
    if (rename(tmp.c_str(), path.c_str()) != 0) throw Error("rename " + tmp + ": " + strerror(errno));
}

static bool path_exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

static void mkdir_p(const std::string &dir)
{
    std::string cur;
    for (size_t i = 0; i <= dir.size(); ++i) {
        if (i == dir.size() || dir[i] == '/') {
            if (!cur.empty() && !path_exists(cur) && mkdir(cur.c_str(), 0755) != 0 && errno != EEXIST)
                throw Error("mkdir " + cur + ": " + strerror(errno));
        }
        if (i < dir.size()) cur.push_back(dir[i]);
    }
}

// ------------------------------------------------------------------ tokenizers
namespace {

void split_whitespace(const std::string &text, std::vector<std::pair<size_t, size_t>> &spans)
{
    size_t i = 0;
    const size_t n = text.size();
    while (i < n) {
        while (i < n && isspace((unsigned char)text[i])) ++i;
        const size_t s = i;
        while (i < n && !isspace((unsigned char)text[i])) ++i;
        if (i > s) spans.emplace_back(s, i - s);
    }
}

class VocabTokenizer : public Tokenizer {
public:
    VocabTokenizer(const std::string &path, const std::string &unk_token)
    {
        const std::string text = read_to_string(path);   // one token per line, id = line index
        std::vector<size_t> lens;
        vocab_.reserve_bytes(text.size());
        lens.reserve(text.size() / 8);
        uint32_t id = 0;
        std::string line;
        for (size_t at = 0; at < text.size();) {
            const void *nl = memchr(text.data() + at, '\n', text.size() - at);
            size_t end = nl ? (size_t)((const char *)nl - text.data()) : text.size();
            size_t len = end - at;
            if (len && text[at + len - 1] == '\r') --len;
            line.assign(text, at, len);
            vocab_.add(line, id++);                       // (a repeated token: the later id wins)
            lens.push_back(len);                          // model2vec-rs takes the median of tk.len(): BYTES, not characters
            at = end + 1;
        }
        vocab_.build();
        size_ = id;
        if (!unk_token.empty()) {
            const int64_t u = vocab_.find(unk_token);
            if (u >= 0) unk_ = (uint32_t)u;
        }
        if (!lens.empty()) {  // model2vec: median length of the vocabulary's tokens
            std::nth_element(lens.begin(), lens.begin() + lens.size() / 2, lens.end());
            median_ = std::max<size_t>(1, lens[lens.size() / 2]);
        }
    }
    void encode(const std::string &text, std::vector<uint32_t> &ids) const override
    {
        std::vector<std::pair<size_t, size_t>> spans;
        split_whitespace(text, spans);
        for (auto &sp : spans) {
            const int64_t hit = vocab_.find(text.data() + sp.first, sp.second);
            if (hit >= 0) ids.push_back((uint32_t)hit);
            else if (unk_) ids.push_back(*unk_);
        }
    }
    std::optional<uint32_t> unk_id() const override { return unk_; }
    size_t median_token_length() const override { return median_; }
    uint64_t vocab_size() const override { return size_; }

private:
    FlatVocab vocab_;
    std::optional<uint32_t> unk_;
    size_t median_ = 5;
    uint64_t size_ = 0;
};

class HashTokenizer : public Tokenizer {
public:
    explicit HashTokenizer(uint64_t v) : v_(v) {}
    void encode(const std::string &text, std::vector<uint32_t> &ids) const override
    {
        std::vector<std::pair<size_t, size_t>> spans;
        split_whitespace(text, spans);
        for (auto &sp : spans)
            ids.push_back((uint32_t)(smt_fnv1a_hash(reinterpret_cast<const uint8_t *>(text.data() + sp.first), sp.second) % v_));
    }
    uint64_t vocab_size() const override { return v_; }

private:
    uint64_t v_;
};

class CallbackTokenizer : public Tokenizer {
public:
    CallbackTokenizer(TokenizeFn fn, uint64_t v, std::optional<uint32_t> unk, size_t median)
        : fn_(std::move(fn)), v_(v), unk_(unk), median_(median) {}
    void encode(const std::string &text, std::vector<uint32_t> &ids) const override { fn_(text, ids); }
    std::optional<uint32_t> unk_id() const override { return unk_; }
    size_t median_token_length() const override { return median_; }
    uint64_t vocab_size() const override { return v_; }

private:
    TokenizeFn fn_;
    uint64_t v_;
    std::optional<uint32_t> unk_;
    size_t median_;
};

}  // namespace

std::unique_ptr<Tokenizer> make_vocab_tokenizer(const std::string &vocab_path, const std::string &unk_token)
{
    return std::make_unique<VocabTokenizer>(vocab_path, unk_token);
}
std::unique_ptr<Tokenizer> make_hash_tokenizer(uint64_t vocab_size) { return std::make_unique<HashTokenizer>(vocab_size); }
std::unique_ptr<Tokenizer> make_callback_tokenizer(TokenizeFn fn, uint64_t vocab_size, std::optional<uint32_t> unk,
                                                   size_t median_len)
{
    return std::make_unique<CallbackTokenizer>(std::move(fn), vocab_size, unk, median_len);
}

// ================================================================== search
namespace search {

StaticModel::StaticModel(smt_group *group, std::unique_ptr<Tokenizer> tok, const float *table, uint64_t V, bool normalize)
    : group_(group), tok_(std::move(tok))
{
    check(smt_sharded_model_create(group, table, V, SMT_DIM, normalize ? 1 : 0, &model_), "StaticModel");
}

StaticModel::StaticModel(smt_group *group, std::unique_ptr<Tokenizer> tok, const std::string &path, uint64_t byte_offset, uint64_t V,
                         bool normalize)
    : group_(group), tok_(std::move(tok))
{
    const char *eager = getenv("SEMTOOLS_EAGER_MODEL");
    if (eager && eager[0] == '1') {
        check(smt_sharded_model_create_from_file(group, path.c_str(), byte_offset, V, SMT_DIM, normalize ? 1 : 0, &model_), "StaticModel");
        return;
    }
    lazy_path_ = path;
    lazy_offset_ = byte_offset;
    lazy_V_ = V;
    lazy_normalize_ = normalize;
    lazy_fd_ = open(path.c_str(), O_RDONLY);
    if (lazy_fd_ < 0) throw Error("cannot open " + path + ": " + strerror(errno));
}

StaticModel::~StaticModel()
{
    if (model_) smt_sharded_model_destroy(model_);
    if (lazy_fd_ >= 0) close(lazy_fd_);
}

smt_sharded_model *StaticModel::full_model() const
{
    if (!model_) {
        check(smt_sharded_model_create_from_file(group_, lazy_path_.c_str(), lazy_offset_, lazy_V_, SMT_DIM, lazy_normalize_ ? 1 : 0, &model_),
              "StaticModel (full table upload)");
        PhaseTimer::mark("model_table_upload");
    }
    return model_;
}

void StaticModel::embed_csr(const std::vector<uint32_t> &ids, const std::vector<uint64_t> &offsets, uint64_t n_lines, float *out_host,
                            smt_sharded_corpus *corpus) const
{
    if (model_ || lazy_fd_ < 0) {
        check(smt_sharded_embed(model_, ids.data(), offsets.data(), n_lines, 0, out_host, corpus, nullptr), "embed");
        return;
    }
    // ---- lazy: which rows does this batch touch?
    if (lazy_slot_.size() != lazy_V_) lazy_slot_.assign(lazy_V_, 0);
    std::vector<uint32_t> uniq;
    for (uint32_t id : ids) {
        if (id >= lazy_V_) throw Error("token id outside the embedding table");
        if (!lazy_slot_[id]) { lazy_slot_[id] = 1; uniq.push_back(id); }
    }
    // a sizeable part of the table, or the 65th small call of this process (`semtools search` over hundreds of files embeds
    // them one by one; each compact table costs a device allocation and an upload): upload all of it, once
    if ((uint64_t)uniq.size() * 16 > lazy_V_ || ++lazy_calls_ > 64) {
        for (uint32_t id : uniq) lazy_slot_[id] = 0;
        check(smt_sharded_embed(full_model(), ids.data(), offsets.data(), n_lines, 0, out_host, corpus, nullptr), "embed");
        return;
    }
    std::sort(uniq.begin(), uniq.end());          // file order: neighbouring rows share pages
    for (size_t s = 0; s < uniq.size(); ++s) lazy_slot_[uniq[s]] = (uint32_t)s + 1;
    std::vector<float> compact(std::max<size_t>(uniq.size(), 1) * SMT_DIM, 0.0f);
    {
        const size_t n_threads = std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), (size_t)8, uniq.size() / 512}));
        std::vector<std::exception_ptr> failed(n_threads);
        auto work = [&](size_t t) {
            try {
                for (size_t s = uniq.size() * t / n_threads; s < uniq.size() * (t + 1) / n_threads; ++s) {
                    const off_t at = (off_t)(lazy_offset_ + (uint64_t)uniq[s] * SMT_DIM * sizeof(float));
                    if (pread(lazy_fd_, &compact[s * SMT_DIM], SMT_DIM * sizeof(float), at) != (ssize_t)(SMT_DIM * sizeof(float)))
                        throw Error("short read from " + lazy_path_);
                }
            } catch (...) { failed[t] = std::current_exception(); }
        };
        if (n_threads == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (size_t t = 0; t < n_threads; ++t) th.emplace_back(work, t);
            for (auto &x : th) x.join();
        }
        for (auto &f : failed) if (f) { for (uint32_t id : uniq) lazy_slot_[id] = 0; std::rethrow_exception(f); }
    }
    std::vector<uint32_t> remapped(ids.size());
    for (size_t i = 0; i < ids.size(); ++i) remapped[i] = lazy_slot_[ids[i]] - 1;
    for (uint32_t id : uniq) lazy_slot_[id] = 0;
    smt_sharded_model *tmp = nullptr;   // (a few MB: replicated like the full table)
    check(smt_sharded_model_create(group_, compact.data(), std::max<size_t>(uniq.size(), 1), SMT_DIM, lazy_normalize_ ? 1 : 0, &tmp), "embed (compact table)");
    const int rc = smt_sharded_embed(tmp, remapped.data(), offsets.data(), n_lines, 0, out_host, corpus, nullptr);
    smt_sharded_model_destroy(tmp);
    check(rc, "embed (compact table)");
}

// ---- phase timer
namespace {
std::vector<std::pair<std::string, double>> g_phases;
std::chrono::steady_clock::time_point g_phase_t0 = std::chrono::steady_clock::now();
}  // namespace
void PhaseTimer::mark(const char *phase)
{
    const auto now = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(now - g_phase_t0).count();
    g_phase_t0 = now;
    for (auto &p : g_phases)
        if (p.first == phase) { p.second += ms; return; }
    g_phases.emplace_back(phase, ms);
}
std::string PhaseTimer::json()
{
    std::string out = "{";
    char buf[64];
    for (size_t i = 0; i < g_phases.size(); ++i) {
        snprintf(buf, sizeof(buf), "%.3f", g_phases[i].second);
        out += (i ? ", \"" : "\"") + g_phases[i].first + "\": " + buf;
    }
    return out + "}";
}

// model2vec-rs truncate_str: keep at most max_tokens * median_token_length characters
static std::string truncate_str(const std::string &s, size_t max_tokens, size_t median_len)
{
    const size_t max_chars = max_tokens * median_len;
    size_t chars = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        if (((unsigned char)s[i] & 0xC0) != 0x80) {
            if (chars == max_chars) return s.substr(0, i);
            ++chars;
        }
    }
    return s;
}

void StaticModel::tokenize_batch(const std::vector<std::string> &sentences, size_t begin, size_t end,
                                 std::optional<size_t> max_length, std::vector<uint32_t> &ids,
                                 std::vector<uint64_t> &offsets) const
{
    // encode_batch_fast is rayon-parallel upstream; here: one slice per hardware thread
    const size_t n = end - begin;
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), n / 256));
    std::vector<std::vector<uint32_t>> part_ids(n_threads);
    std::vector<std::vector<uint64_t>> part_len(n_threads);
    const auto unk = tok_->unk_id();
    // a throwing tokenizer (callback failure, bad_alloc) must reach the caller -- and through it the extern "C"
    // wrappers' catch blocks -- not std::terminate the process from a worker thread
    std::vector<std::exception_ptr> failed(n_threads);
    auto work = [&](size_t t) {
      try {
        const size_t b = begin + n * t / n_threads, e = begin + n * (t + 1) / n_threads;
        std::vector<uint32_t> tmp;
        for (size_t i = b; i < e; ++i) {
            tmp.clear();
            if (max_length) tok_->encode(truncate_str(sentences[i], *max_length, tok_->median_token_length()), tmp);
            else tok_->encode(sentences[i], tmp);
            if (unk) tmp.erase(std::remove(tmp.begin(), tmp.end(), *unk), tmp.end());
            if (max_length && tmp.size() > *max_length) tmp.resize(*max_length);
            part_ids[t].insert(part_ids[t].end(), tmp.begin(), tmp.end());
            part_len[t].push_back(tmp.size());
        }
      } catch (...) { failed[t] = std::current_exception(); }
    };
    if (n_threads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < n_threads; ++t) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    for (auto &f : failed) if (f) std::rethrow_exception(f);
    ids.clear();
    offsets.assign(1, 0);
    for (size_t t = 0; t < n_threads; ++t) {
        ids.insert(ids.end(), part_ids[t].begin(), part_ids[t].end());
        for (uint64_t l : part_len[t]) offsets.push_back(offsets.back() + l);
    }
}

std::vector<std::vector<float>> StaticModel::encode_with_args(const std::vector<std::string> &sentences,
                                                              std::optional<size_t> max_length,
                                                              size_t batch_size) const
{
    std::vector<std::vector<float>> out;
    out.reserve(sentences.size());
    std::vector<uint32_t> ids;
    std::vector<uint64_t> offsets;
    std::vector<float> buf;
    if (batch_size == 0) batch_size = 1;
    for (size_t b = 0; b < sentences.size(); b += batch_size) {
        const size_t e = std::min(sentences.size(), b + batch_size);
        tokenize_batch(sentences, b, e, max_length, ids, offsets);
        buf.resize((e - b) * SMT_DIM);
        embed_csr(ids, offsets, e - b, buf.data(), nullptr);   // (tokenize_batch already truncated to max_length)
        for (size_t i = 0; i < e - b; ++i) out.emplace_back(buf.begin() + i * SMT_DIM, buf.begin() + (i + 1) * SMT_DIM);
    }
    return out;
}

uint64_t StaticModel::encode_into(const std::vector<std::string> &sentences, std::optional<size_t> max_length,
                                  size_t batch_size, smt_sharded_corpus *corpus, TokenCsr *sink) const
{
    // Double-buffered pipeline: while the GPU gathers/pools batch i (H2D of the ids + K1),
    // the host threads already tokenise batch i+1.  Batches are appended in order, so rows == line order.
    const uint64_t first = smt_sharded_corpus_rows(corpus);
    if (batch_size == 0) batch_size = 1;
    // The reference's 16384-line batches are an allocation bound of its own pipeline; rows do not depend on how the lines
    // are batched.  Here a batch is what one round of tokenizer threads chews on, so it must be large enough to amortise
    // starting them: SEMTOOLS_EMBED_BATCH overrides (default: the caller's value, at least 65536).
    {
        static const size_t env_batch = [] { const char *e = getenv("SEMTOOLS_EMBED_BATCH"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)0; }();
        batch_size = env_batch ? env_batch : std::max<size_t>(batch_size, 65536);
    }
    struct Slot { std::vector<uint32_t> ids; std::vector<uint64_t> offsets; };
    Slot slots[2];
    const size_t n = sentences.size();
    if (n == 0) return first;
    if (!model_ && n > 32768) full_model();   // lazy mode: a large job wants the whole table, decide before the pipeline starts
    auto tokenize = [&](size_t b, Slot &s) { tokenize_batch(sentences, b, std::min(n, b + batch_size), max_length, s.ids, s.offsets); };
    tokenize(0, slots[0]);
    int cur = 0;
    for (size_t b = 0; b < n; b += batch_size) {
        const size_t e = std::min(n, b + batch_size);
        std::thread next;
        std::exception_ptr next_failed;
        if (e < n) next = std::thread([&, e, cur]() {
            try { tokenize(e, slots[cur ^ 1]); } catch (...) { next_failed = std::current_exception(); }
        });
        std::exception_ptr embed_failed;
        try { embed_csr(slots[cur].ids, slots[cur].offsets, e - b, nullptr, corpus); } catch (...) { embed_failed = std::current_exception(); }
        if (next.joinable()) next.join();
        if (embed_failed) std::rethrow_exception(embed_failed);
        if (next_failed) std::rethrow_exception(next_failed);
        if (sink) {  // (tokenize_batch already dropped unk ids and truncated: these are exactly the ids that were pooled)
            sink->ids.insert(sink->ids.end(), slots[cur].ids.begin(), slots[cur].ids.end());
            for (size_t i = 0; i + 1 < slots[cur].offsets.size(); ++i)
                sink->lens.push_back((uint32_t)(slots[cur].offsets[i + 1] - slots[cur].offsets[i]));
        }
        cur ^= 1;
    }
    return first;
}

void StaticModel::embed_tokens_into(const uint32_t *ids, const uint64_t *offsets, uint64_t n_lines, smt_sharded_corpus *corpus) const
{
    check(smt_sharded_embed(full_model(), ids, offsets, n_lines, 0, nullptr, corpus, nullptr), "embed_tokens_into");
}

uint64_t StaticModel::tokenizer_fingerprint() const
{
    // FNV-1a over: vocab size, unk id, median token length, and the ids of a probe text that exercises casing,
    // accents, digits, punctuation and an unknown-looking word
    uint64_t hsh = 0xcbf29ce484222325ull;
    auto mix = [&](uint64_t v) { for (int i = 0; i < 8; ++i) { hsh ^= (v >> (8 * i)) & 0xFF; hsh *= 0x100000001b3ull; } };
    mix(tok_->vocab_size());
    mix(tok_->unk_id() ? (uint64_t)*tok_->unk_id() + 1 : 0);
    mix(tok_->median_token_length());
    std::vector<uint32_t> ids;
    tok_->encode("The quick brown fox, jumping over 12 lazy dogs: caf\xc3\xa9 na\xc3\xaf" "ve Stra\xc3\x9f" "e _foo-bar_ w17 zqxjkv!", ids);
    mix(ids.size());
    for (uint32_t id : ids) mix(id);
    return hsh ? hsh : 1;
}

std::vector<float> StaticModel::encode_single(const std::string &sentence) const
{
    // StaticModel::encode(&[s]) == encode_with_args(.., Some(512), 1024)  [UPSTREAM-RECALL]
    return encode_with_args({sentence}, 512, 1024).at(0);
}

Embeddings::Embeddings(smt_group *group) { check(smt_sharded_corpus_create(group, SMT_DIM, &corpus_), "Embeddings"); }
Embeddings::~Embeddings() { smt_sharded_corpus_destroy(corpus_); }
uint64_t Embeddings::rows() const { return smt_sharded_corpus_rows(corpus_); }

std::optional<Document> create_document_from_content(const std::string &filename, const std::string &content,
                                                     const StaticModel &model, bool ignore_case, Embeddings &emb)
{
    std::vector<std::string> lines = lines_of(content);
    if (lines.empty()) return std::nullopt;  // mod.rs:57-59
    PhaseTimer::mark("split_lines");
    Document doc;
    doc.filename = filename;
    if (ignore_case) {
        std::vector<std::string> lowered;
        lowered.reserve(lines.size());
        for (auto &s : lines) lowered.push_back(to_lowercase(s));
        doc.first_row = model.encode_into(lowered, 2048, 16384, emb.corpus());  // mod.rs:69
    } else {
        doc.first_row = model.encode_into(lines, 2048, 16384, emb.corpus());
    }
    PhaseTimer::mark("tokenize_and_embed");
    doc.lines = std::move(lines);
    return doc;
}

std::vector<SearchResult> search_documents(const std::vector<Document> &documents, const Embeddings &emb,
                                           const std::vector<float> &query_embedding, const SearchConfig &config)
{
    if (query_embedding.size() != SMT_DIM) return {};  // f32::cosine -> None on length mismatch: every row skipped
    return std::move(search_documents_batch(documents, emb, {query_embedding}, config).at(0));
}

std::vector<std::vector<SearchResult>> search_documents_batch(const std::vector<Document> &documents, const Embeddings &emb,
                                                              const std::vector<std::vector<float>> &query_embeddings,
                                                              const SearchConfig &config)
{
    const size_t nq = query_embeddings.size();
    std::vector<std::vector<SearchResult>> all(nq);
    if (documents.empty() || nq == 0) return all;
    std::vector<float> qflat(nq * SMT_DIM);
    for (size_t q = 0; q < nq; ++q) {
        if (query_embeddings[q].size() != SMT_DIM) throw Error("query embedding must have 256 dimensions");
        std::copy(query_embeddings[q].begin(), query_embeddings[q].end(), qflat.begin() + q * SMT_DIM);
    }

    // one row range per document, in slice order (== the reference's nested loop order)
    std::vector<smt_range> ranges;
    std::vector<uint64_t> starts;  // first row of each document
    uint64_t total = 0;
    for (auto &d : documents) {
        if (!starts.empty() && d.first_row < starts.back())
            throw Error("search_documents: documents must be in embedding order");
        starts.push_back(d.first_row);
        const uint64_t end = d.first_row + d.lines.size();
        if (!ranges.empty() && ranges.back().end == d.first_row) ranges.back().end = end;
        else ranges.push_back({d.first_row, end});
        total += d.lines.size();
    }
    if (total == 0) return all;
    // the whole corpus in order is the common case: no filter needed (lets batches take the MFMA path)
    const bool whole = ranges.size() == 1 && ranges[0].begin == 0 && ranges[0].end == emb.rows();

    const bool all_hits = config.max_distance.has_value();
    if (!all_hits && config.top_k == 0) return all;
    uint64_t cap = all_hits ? std::min<uint64_t>(total, 4096) : std::min<uint64_t>(config.top_k, total);
    std::vector<uint64_t> rows, counts(nq);
    std::vector<double> dist;
    for (;;) {
        rows.resize(cap * nq);
        dist.resize(cap * nq);
        const int rc = smt_sharded_search(emb.corpus(), qflat.data(), (uint32_t)nq, (uint32_t)std::min<size_t>(config.top_k, 0xFFFFFFFFu),
                                          all_hits ? *config.max_distance : NAN, SMT_MODE_DOCUMENTS, whole ? nullptr : ranges.data(),
                                          whole ? 0 : (uint32_t)ranges.size(), rows.data(), dist.data(), counts.data(), cap);
        if (rc == SMT_E_TRUNCATED) { cap = *std::max_element(counts.begin(), counts.end()); continue; }
        check(rc, "search_documents");
        break;
    }
    for (size_t q = 0; q < nq; ++q) {
        std::vector<SearchResult> &results = all[q];
        results.reserve(counts[q]);
        for (uint64_t i = 0; i < counts[q]; ++i) {
            const uint64_t row = rows[q * cap + i];
            const size_t di = (size_t)(std::upper_bound(starts.begin(), starts.end(), row) - starts.begin()) - 1;
            const Document &doc = documents[di];
            const size_t idx = (size_t)(row - doc.first_row);
            const size_t bottom = idx > config.n_lines ? idx - config.n_lines : 0;   // saturating_sub  (mod.rs:90)
            const size_t top = std::min(doc.lines.size(), idx + config.n_lines + 1);  // (mod.rs:91)
            SearchResult r;
            r.filename = doc.filename;
            r.lines.assign(doc.lines.begin() + bottom, doc.lines.begin() + top);
            r.distance = dist[q * cap + i];
            r.start = bottom;
            r.end = top;
            r.match_line = idx;
            results.push_back(std::move(r));
        }
    }
    return all;  // each already (distance asc, document/line order) == stable sort; take(top_k) done on device
}

std::vector<SearchResult> search_files(const std::vector<std::string> &files, const std::string &query,
                                       const StaticModel &model, const SearchConfig &config)
{
    PhaseTimer::mark("model_load");
    Embeddings emb(model.group());
    std::vector<Document> documents;
    // mod.rs:128-134 reads and embeds file by file; the rows do not depend on how the lines are batched, so all files
    // are read first (the first error still aborts before anything is printed) and embedded in ONE pipeline run
    // (tokenise || H2D || K1 across file borders instead of a round of threads and a K1 launch per file)
    std::vector<std::string> all;
    for (auto &f : files) {
        const std::string content = read_to_string(f);  // `?`: first error aborts (mod.rs:130)
        std::vector<std::string> lines = lines_of(content);
        if (lines.empty()) continue;                     // create_document_from_content -> None (mod.rs:57-59)
        Document doc;
        doc.filename = f;
        doc.first_row = all.size();
        for (auto &l : lines) all.push_back(config.ignore_case ? to_lowercase(l) : l);   // mod.rs:61-67: embed the lowered copy
        doc.lines = std::move(lines);
        documents.push_back(std::move(doc));
    }
    PhaseTimer::mark("split_lines");
    if (!all.empty()) {
        const uint64_t first = model.encode_into(all, 2048, 16384, emb.corpus());     // mod.rs:69
        for (auto &d : documents) d.first_row += first;
    }
    PhaseTimer::mark("tokenize_and_embed");
    const std::vector<float> query_embedding = model.encode_single(query);
    PhaseTimer::mark("embed_query");
    auto res = search_documents(documents, emb, query_embedding, config);
    PhaseTimer::mark("scan_select");
    return res;
}

std::vector<workspace::RankedLine> search_with_workspace(const std::vector<std::string> &files, const std::string &query,
                                                         const StaticModel &model, const SearchConfig &config,
                                                         const std::optional<std::string> &workspace_name)
{
    using namespace workspace;
    PhaseTimer::mark("model_load");
    const std::vector<float> query_embedding = model.encode_single(query);
    PhaseTimer::mark("embed_query");
    Workspace ws = Workspace::open(workspace_name);
    auto store = Store::open(ws.config.root_dir, model.group());
    PhaseTimer::mark("store_open_corpus_load");
    {
        // the approximate index is OPT-IN: the reference's store always searches exactly (store.rs:619,632)
        const char *min_rows = getenv("SEMTOOLS_INDEX_MIN_ROWS"), *nprobe = getenv("SEMTOOLS_INDEX_NPROBE");
        uint64_t min = ws.config.approximate_index_min_rows ? ws.config.approximate_index_min_rows : UINT64_MAX;
        if (min_rows) { min = strtoull(min_rows, nullptr, 10); if (min == 0) min = UINT64_MAX; }
        store->set_index_policy(ws.config.oversample_factor, min, nprobe ? (uint32_t)strtoul(nprobe, nullptr, 10) : 16u);
    }

    // Step 1: changed / new / unchanged (mod.rs:158)
    const std::vector<DocumentState> doc_states = store->analyze_document_states(files);

    PhaseTimer::mark("change_detection");
    // Step 2+3: embed new/changed documents straight into the resident store
    size_t n_lines_upserted = 0;
    std::vector<DocMeta> docs_to_upsert;
    std::vector<std::pair<std::string, std::vector<std::string>>> pending;
    for (auto &st : doc_states) {
        if (st.kind == DocumentState::Unchanged) continue;
        std::vector<std::string> lines = lines_of(st.info.content);
        if (lines.empty()) continue;  // create_document_from_content -> None
        if (config.ignore_case) for (auto &s : lines) s = to_lowercase(s);
        n_lines_upserted += lines.size();
        pending.emplace_back(st.info.filename, std::move(lines));
        docs_to_upsert.push_back(st.info.meta);
    }
    if (n_lines_upserted) {
        fprintf(stderr, "Updating workspace with %zu lines from new/changed docs...\n", n_lines_upserted);  // mod.rs:194-197
        store->upsert_documents_lines(pending, model);
        store->compact_if_sparse();  // every re-embedded document left its old rows behind: bound the dead rows
        store->flush_line_embeddings();
    }
    if (!docs_to_upsert.empty()) {
        fprintf(stderr, "Updating workspace with %zu new/changed documents...\n", docs_to_upsert.size());  // mod.rs:203-206
        store->upsert_document_metadata(docs_to_upsert);
    }

    PhaseTimer::mark("embed_and_persist_changed_files");
    // Step 4 (mod.rs:211-213)
    std::optional<float> max_distance;
    if (config.max_distance) max_distance = (float)*config.max_distance;
    auto ranked = store->search_line_embeddings(query_embedding, files, config.top_k, max_distance);
    PhaseTimer::mark("scan_select");
    return ranked;
}

}  // namespace search

// ================================================================== workspace
namespace workspace {

static std::string home_dir()
{
    const char *h = getenv("HOME");
    if (!h || !*h) throw Error("No home dir found?");
    return h;
}

std::string Workspace::root_path(const std::string &name) { return home_dir() + "/.semtools/workspaces/" + name; }
std::string Workspace::config_path_for(const std::string &name) { return root_path(name) + "/config.json"; }

std::string Workspace::active(const std::optional<std::string> &workspace_name)
{
    std::string a;
    if (workspace_name) a = *workspace_name;
    else if (const char *e = getenv("SEMTOOLS_WORKSPACE")) a = e;
    if (a.empty()) throw Error("No active workspace. Run: workspace use <name>");
    return a;
}

std::string Workspace::active_path(const std::optional<std::string> &workspace_name) { return root_path(active(workspace_name)); }

Workspace Workspace::open(const std::optional<std::string> &workspace_name)
{
    const std::string act = active(workspace_name);
    Workspace ws;
    try {
        const json::Value v = json::parse(read_to_string(config_path_for(act)));
        if (auto *x = v.get("name")) ws.config.name = x->s;
        if (auto *x = v.get("root_dir")) ws.config.root_dir = x->s;
        if (auto *x = v.get("in_batch_size")) ws.config.in_batch_size = (size_t)x->as_u64();
        if (auto *x = v.get("oversample_factor")) ws.config.oversample_factor = (size_t)x->as_u64();
        if (auto *x = v.get("approximate_index_min_rows")) ws.config.approximate_index_min_rows = x->as_u64();
    } catch (const std::exception &) {
        ws.config = WorkspaceConfig();  // unreadable / invalid config -> defaults (mod.rs:36-40)
    }
    if (ws.config.root_dir.empty()) ws.config.root_dir = root_path(act);
    if (ws.config.name.empty() || ws.config.name == "default") ws.config.name = act;
    return ws;
}

void Workspace::save() const
{
    const std::string path = config_path_for(config.name);
    mkdir_p(path.substr(0, path.rfind('/')));
    json::Value v = json::Value::object();
    v.set("name", json::Value::str(config.name));
    v.set("root_dir", json::Value::str(config.root_dir));
    v.set("in_batch_size", json::Value::uint(config.in_batch_size));
    v.set("oversample_factor", json::Value::uint(config.oversample_factor));
    if (config.approximate_index_min_rows) v.set("approximate_index_min_rows", json::Value::uint(config.approximate_index_min_rows));
    write_file_atomic(path, json::to_string_pretty(v));
}

uint64_t DocMeta::id() const { return smt_doc_meta_id(path.c_str()); }
uint64_t LineEmbedding::id() const { return smt_line_embedding_id(path.c_str(), line_number); }

std::unique_ptr<Store> Store::open(const std::string &workspace_dir, smt_group *group)
{
    std::unique_ptr<Store> s(new Store());
    s->dir_ = workspace_dir;
    s->group_ = group;
    int n_ranks = 1;
    check(smt_group_info(group, &n_ranks, nullptr, nullptr, nullptr, nullptr), "Store::open");
    mkdir_p(workspace_dir);
    const std::string docs = workspace_dir + "/documents.json";
    const std::string rows = workspace_dir + "/line_rows.json";
    const std::string emb = workspace_dir + "/line_embeddings.f32";
    if (path_exists(docs)) {
        const json::Value v = json::parse(read_to_string(docs));
        for (auto &d : v.arr) {
            DocMeta m;
            if (auto *x = d.get("path")) m.path = x->s;
            if (auto *x = d.get("size_bytes")) m.size_bytes = x->as_u64();
            if (auto *x = d.get("mtime")) m.mtime = x->as_i64();
            // a store written before `_version` existed counts as version 1 (store.rs:32-33)
            m._version = 1;
            if (auto *x = d.get("_version")) m._version = (uint32_t)x->as_u64();
            s->docs_[m.path] = m;
        }
    }
    bool corpus_ok = false;
    json::Value rows_json;
    bool have_rows_json = false;
    if (path_exists(rows)) {
        try { rows_json = json::parse(read_to_string(rows)); have_rows_json = true; } catch (const std::exception &) {}
    }
    // how the rows were dealt over the GPUs when the store was written (see flush_line_embeddings): restored when this
    // group has as many ranks, so that shards -- and the per-shard index files -- are what they were
    std::vector<uint64_t> piece_rows;
    std::vector<uint32_t> piece_rank;
    if (have_rows_json && n_ranks > 1)
        if (auto *sh = rows_json.get("shards"))
            if (auto *nr = sh->get("n_ranks"); nr && nr->as_u64() == (uint64_t)n_ranks)
                if (auto *pc = sh->get("pieces"))
                    for (auto &e : pc->arr)
                        if (e.arr.size() == 2) { piece_rows.push_back(e.arr[0].as_u64()); piece_rank.push_back((uint32_t)e.arr[1].as_u64()); }
    bool layout_restored = false;
    if (path_exists(emb)) {
        // A truncated / foreign file must not brick the workspace: start from an empty store, every document then
        // counts as Changed (no extent, see analyze_document_states) and is re-embedded by the next search.
        int rc = SMT_E_INVALID;
        if (!piece_rows.empty()) {
            rc = smt_sharded_corpus_load_layout(group, emb.c_str(), piece_rows.data(), piece_rank.data(), piece_rows.size(), &s->corpus_);
            layout_restored = rc == SMT_OK;
        }
        if (rc == SMT_E_INVALID) rc = smt_sharded_corpus_load(group, emb.c_str(), &s->corpus_);   // (stale layout: cut evenly)
        if (rc == SMT_OK) {
            s->rows_on_disk_ = smt_sharded_corpus_rows(s->corpus_);
            s->rows_on_disk_valid_ = true;
            corpus_ok = true;
        } else if (rc != SMT_E_IO) {
            check(rc, "Store::open");
        } else {
            fprintf(stderr, "warning: %s is unreadable (%s); the workspace will be re-embedded\n", emb.c_str(), smt_last_error());
        }
    }
    if (!corpus_ok) check(smt_sharded_corpus_create(group, SMT_DIM, &s->corpus_), "Store::open");
    // the index files name LOCAL rows by position: only valid on the layout they were built on
    s->index_on_disk_ = corpus_ok && (n_ranks == 1 || layout_restored) && path_exists(s->index_file(0));
    if (corpus_ok && have_rows_json) {
        const json::Value &v = rows_json;
        uint64_t live = 0;
        if (auto *arr = v.get("extents"))
            for (auto &e : arr->arr) {
                Extent x;
                x.first_row = e.get("first_row")->as_u64();
                x.n_rows = e.get("n_rows")->as_u64();
                // torn write: drop the extent; analyze_document_states reports a document without one as Changed
                if (x.first_row + x.n_rows > smt_sharded_corpus_rows(s->corpus_)) continue;
                s->extents_[e.get("path")->s] = x;
                live += x.n_rows;
            }
        s->dead_rows_ = smt_sharded_corpus_rows(s->corpus_) - live;
    }
    return s;
}

Store::~Store()
{
    if (token_log_file_) fclose(token_log_file_);
    if (index_) smt_sharded_ivfpq_destroy(index_);  // (before the corpus it points into)
    smt_sharded_corpus_destroy(corpus_);
}

// rank r's part of the index: `line_index.ivf` on one GPU, `line_index.ivf.r<r>of<n>` on several (smt_sharded_ivfpq_save)
std::string Store::index_file(int rank) const
{
    int n_ranks = 1;
    (void)smt_group_info(group_, &n_ranks, nullptr, nullptr, nullptr, nullptr);
    const std::string base = dir_ + "/line_index.ivf";
    return n_ranks == 1 ? base : base + ".r" + std::to_string(rank) + "of" + std::to_string(n_ranks);
}

// The index points into corpus_ and names its rows by position: it goes BEFORE the corpus is destroyed or its rows move
// (smt_ivfpq_destroy reads index->corpus->ctx: the other order is a use-after-free).
void Store::drop_index()
{
    if (index_) { smt_sharded_ivfpq_destroy(index_); index_ = nullptr; }
    if (index_on_disk_) {
        int n_ranks = 1;
        (void)smt_group_info(group_, &n_ranks, nullptr, nullptr, nullptr, nullptr);
        for (int r = 0; r < n_ranks; ++r) (void)remove(index_file(r).c_str());
        index_on_disk_ = false;
    }
}

void Store::set_index_policy(size_t oversample_factor, uint64_t min_rows, uint32_t nprobe)
{
    oversample_factor_ = std::max<size_t>(1, oversample_factor);
    index_min_rows_ = min_rows;
    index_nprobe_ = std::max<uint32_t>(1, nprobe);
}

// Bring the approximate index in line with the corpus: load it from disk, extend it by the rows appended since, or
// (re)build it.  Returns false when no usable index exists (then the caller scans exactly).
bool Store::ensure_index() const
{
    const uint64_t rows = smt_sharded_corpus_rows(corpus_);
    const std::string file = dir_ + "/line_index.ivf";
    int n_ranks = 1;
    (void)smt_group_info(group_, &n_ranks, nullptr, nullptr, nullptr, nullptr);
    bool changed = false;
    if (!index_ && index_on_disk_) {
        if (smt_sharded_ivfpq_load(corpus_, file.c_str(), &index_) != SMT_OK) index_ = nullptr;  // stale / corrupt: rebuild below
        if (index_) { uint64_t n = 0; smt_sharded_ivfpq_info(index_, &n, nullptr, nullptr); index_built_rows_ = n; }
    }
    if (index_) {
        uint64_t covered = 0;
        smt_sharded_ivfpq_info(index_, &covered, nullptr, nullptr);
        if (covered < rows) {
            // the corpus only grew: incremental insert with the existing quantisers -- until it has doubled
            uint64_t added = 0;
            if (rows > 2 * std::max<uint64_t>(index_built_rows_, 1) || smt_sharded_ivfpq_append(index_, &added) != SMT_OK) {
                smt_sharded_ivfpq_destroy(index_);
                index_ = nullptr;
            } else {
                changed = true;
            }
        }
    }
    if (!index_) {
        smt_ivfpq_params prm;
        memset(&prm, 0, sizeof(prm));
        // ~ sqrt(N) lists per shard (a multiple of 32 in [32, 4096]): 10 M rows -> 4096 lists of ~2.4 k rows
        std::vector<uint64_t> per_rank((size_t)n_ranks, 0);
        (void)smt_sharded_corpus_rank_rows(corpus_, per_rank.data());
        const uint64_t smallest = *std::min_element(per_rank.begin(), per_rank.end());
        uint64_t nlist = (uint64_t)std::sqrt((double)(rows / (uint64_t)n_ranks)) * 4 / 3;
        nlist = std::min<uint64_t>(4096, std::max<uint64_t>(32, nlist / 32 * 32));
        prm.nlist = (uint32_t)nlist;
        prm.m = 32;
        prm.nbits = 8;
        prm.train_iters = 10;
        prm.local_pca = 1;
        // several GPUs: ONE set of nlist lists over the whole corpus (centroid sums all-reduced in the k-means), each list
        // spread over the shards; quantisers and codes are fitted per shard
        if (smallest < nlist || smt_sharded_ivfpq_build(corpus_, &prm, n_ranks > 1 ? 1 : 0, &index_) != SMT_OK) { index_ = nullptr; return false; }
        index_built_rows_ = rows;
        changed = true;
    }
    if (changed) {  // persist beside the vectors (a sibling first, then rename: smt_sharded_ivfpq_save)
        if (smt_sharded_ivfpq_save(index_, file.c_str()) == SMT_OK) index_on_disk_ = true;
    }
    return true;
}

std::unordered_map<std::string, DocMeta> Store::get_existing_docs(const std::vector<std::string> &paths) const
{
    std::unordered_map<std::string, DocMeta> out;
    for (auto &p : paths) {
        auto it = docs_.find(p);
        if (it != docs_.end()) out[p] = it->second;
    }
    return out;
}

void Store::delete_document_metadata(const std::vector<std::string> &paths)
{
    if (paths.empty()) return;
    for (auto &p : paths) {
        auto it = docs_.find(p);
        // the reference's delete filter also requires _version == CURRENT (store.rs:262-269)
        if (it != docs_.end() && it->second._version == CURRENT_EMBEDDING_VERSION) docs_.erase(it);
    }
    flush_documents();
}

void Store::delete_line_embeddings(const std::vector<std::string> &paths)
{
    if (paths.empty()) return;
    for (auto &p : paths) {
        auto it = extents_.find(p);
        if (it != extents_.end()) {
            dead_rows_ += it->second.n_rows;
            extents_.erase(it);
            token_log_append(p, nullptr, 0);   // tombstone (no-op without a log)
        }
    }
    compact_if_sparse();
    flush_line_embeddings();
}

void Store::delete_documents(const std::vector<std::string> &paths)
{
    if (paths.empty()) return;
    delete_document_metadata(paths);
    delete_line_embeddings(paths);
}

void Store::upsert_document_metadata(const std::vector<DocMeta> &metas)
{
    if (metas.empty()) return;
    for (auto &m : metas) docs_[m.path] = m;  // same path -> same id -> replacement (store.rs:951-1000)
    flush_documents();
}

void Store::upsert_line_embeddings(const std::vector<LineEmbedding> &line_embeddings)
{
    if (line_embeddings.empty()) return;
    // group by path (ids are per (path, line): an upsert replaces that line, adds it if new)
    std::map<std::string, std::vector<const LineEmbedding *>> by_path;
    for (auto &le : line_embeddings) {
        if (le.embedding.size() != LINE_EMBEDDING_SIZE) throw Error("line embedding must have 256 dimensions");
        if (le.line_number < 0) throw Error("negative line_number");
        by_path[le.path].push_back(&le);
    }
    for (auto &kv : by_path) {
        uint64_t max_line = 0;
        for (auto *le : kv.second) max_line = std::max<uint64_t>(max_line, (uint64_t)le->line_number);
        auto it = extents_.find(kv.first);
        const uint64_t old_n = it != extents_.end() ? it->second.n_rows : 0;
        if (it != extents_.end() && max_line < old_n) {  // in-place replacement
            for (auto *le : kv.second)
                check(smt_sharded_corpus_write_rows(corpus_, it->second.first_row + (uint64_t)le->line_number, le->embedding.data(), 1),
                      "upsert_line_embeddings");
            rows_on_disk_valid_ = false;  // rows already on disk changed: the next flush rewrites the file
            drop_index();                 // ... and the index's codes of those rows are stale
            continue;
        }
        const uint64_t new_n = std::max(old_n, max_line + 1);
        std::vector<float> rows((size_t)new_n * LINE_EMBEDDING_SIZE, 0.0f);
        if (old_n) check(smt_sharded_corpus_read_rows(corpus_, it->second.first_row, old_n, rows.data()), "upsert_line_embeddings");
        for (auto *le : kv.second)
            std::copy(le->embedding.begin(), le->embedding.end(), rows.begin() + (size_t)le->line_number * LINE_EMBEDDING_SIZE);
        uint64_t first = 0;
        check(smt_sharded_corpus_append_host(corpus_, rows.data(), new_n, &first), "upsert_line_embeddings");
        dead_rows_ += old_n;
        extents_[kv.first] = Extent{first, new_n};
    }
    for (auto &kv : by_path) token_log_append(kv.first, nullptr, 0);   // vectors from outside: no tokens are known for them
    compact_if_sparse();
    flush_line_embeddings();
}

static bool token_cache_enabled();

void Store::upsert_documents_lines(std::vector<std::pair<std::string, std::vector<std::string>>> &docs, const search::StaticModel &model)
{
    if (docs.empty()) return;
    if (docs.size() == 1) { upsert_document_lines(docs[0].first, docs[0].second, model); return; }
    size_t total = 0;
    for (auto &d : docs) total += d.second.size();
    std::vector<std::string> all;
    all.reserve(total);
    for (auto &d : docs) {
        auto it = extents_.find(d.first);
        if (it != extents_.end()) dead_rows_ += it->second.n_rows;  // the whole old document is replaced (no stale tail)
        for (auto &l : d.second) all.push_back(std::move(l));
    }
    const bool cache = token_cache_enabled();
    search::TokenCsr tokens;
    uint64_t row = model.encode_into(all, 2048, 16384, corpus_, cache ? &tokens : nullptr);
    const uint64_t fingerprint = cache ? model.tokenizer_fingerprint() : 0;
    size_t line = 0, id_at = 0;
    for (auto &d : docs) {
        const size_t n = d.second.size();
        extents_[d.first] = Extent{row, (uint64_t)n};
        if (cache) {
            search::TokenCsr one;
            one.lens.assign(tokens.lens.begin() + line, tokens.lens.begin() + line + n);
            size_t n_ids = 0;
            for (uint32_t l : one.lens) n_ids += l;
            one.ids.assign(tokens.ids.begin() + id_at, tokens.ids.begin() + id_at + n_ids);
            token_log_append(d.first, &one, fingerprint);
            id_at += n_ids;
        }
        row += n;
        line += n;
    }
}

void Store::upsert_document_lines(const std::string &path, const std::vector<std::string> &lines_for_embedding,
                                  const search::StaticModel &model)
{
    auto it = extents_.find(path);
    if (it != extents_.end()) dead_rows_ += it->second.n_rows;  // the whole old document is replaced (no stale tail)
    const bool cache = token_cache_enabled();
    search::TokenCsr tokens;
    const uint64_t first = model.encode_into(lines_for_embedding, 2048, 16384, corpus_, cache ? &tokens : nullptr);
    extents_[path] = Extent{first, (uint64_t)lines_for_embedding.size()};
    if (cache) token_log_append(path, &tokens, model.tokenizer_fingerprint());
}

// ---- token cache: <dir>/line_tokens.log
//   header  : "SMTTOK01", u64 tokenizer fingerprint, u64 reserved
//   record  : u32 'TOKD', u32 path bytes, u32 n_lines (0xFFFFFFFF = tombstone), u32 reserved, u64 n_ids,
//             path, u32 lens[n_lines], u32 ids[n_ids]
// Append-only; the latest record of a path wins; a torn tail record is ignored by the reader.
static bool token_cache_enabled()
{
    const char *e = getenv("SEMTOOLS_TOKEN_CACHE");
    return !(e && e[0] == '0');
}
static constexpr uint32_t TOK_TAG = 0x444B4F54u, TOK_TOMBSTONE = 0xFFFFFFFFu;

void Store::token_log_append(const std::string &path, const search::TokenCsr *tokens, uint64_t fingerprint) const
{
    const std::string log = dir_ + "/line_tokens.log";
    if (token_log_fingerprint_ == 0 && path_exists(log)) {
        FILE *f = fopen(log.c_str(), "rb");
        char magic[8];
        uint64_t fp = 0;
        if (f && fread(magic, 1, 8, f) == 8 && memcmp(magic, "SMTTOK01", 8) == 0 && fread(&fp, 8, 1, f) == 1) token_log_fingerprint_ = fp;
        if (f) fclose(f);
    }
    if (!tokens && token_log_fingerprint_ == 0) return;  // tombstone into a log that does not exist: nothing to cancel
    if (tokens && fingerprint == 0) return;
    const bool fresh = tokens && token_log_fingerprint_ != fingerprint;   // no log yet, or tokens of ANOTHER tokenizer: start over
    // the stream stays open over a series of appends (a repository is thousands of small files) and is closed -- i.e.
    // flushed -- by token_log_close(), which flush_line_embeddings and the destructor call
    if (fresh && token_log_file_) { fclose(token_log_file_); token_log_file_ = nullptr; }
    if (!token_log_file_) token_log_file_ = fopen(log.c_str(), fresh ? "wb" : "ab");
    FILE *f = token_log_file_;
    if (!f) throw Error("cannot open " + log + ": " + strerror(errno));
    bool ok = true;
    if (fresh) {
        const uint64_t zero = 0;
        ok = fwrite("SMTTOK01", 1, 8, f) == 8 && fwrite(&fingerprint, 8, 1, f) == 1 && fwrite(&zero, 8, 1, f) == 1;
        token_log_fingerprint_ = fingerprint;
    }
    const uint32_t head[4] = {TOK_TAG, (uint32_t)path.size(), tokens ? (uint32_t)tokens->lens.size() : TOK_TOMBSTONE, 0};
    const uint64_t n_ids = tokens ? tokens->ids.size() : 0;
    ok = ok && fwrite(head, 4, 4, f) == 4 && fwrite(&n_ids, 8, 1, f) == 1 && fwrite(path.data(), 1, path.size(), f) == path.size();
    if (tokens && !tokens->lens.empty()) ok = ok && fwrite(tokens->lens.data(), 4, tokens->lens.size(), f) == tokens->lens.size();
    if (n_ids) ok = ok && fwrite(tokens->ids.data(), 4, n_ids, f) == n_ids;
    if (!ok) { token_log_close(); throw Error("short write to " + log); }
}

void Store::token_log_close() const
{
    if (!token_log_file_) return;
    const bool ok = fclose(token_log_file_) == 0;
    token_log_file_ = nullptr;
    if (!ok) throw Error("short write to " + dir_ + "/line_tokens.log");
}

Store::ReembedReport Store::reembed_from_token_cache(const search::StaticModel &model)
{
    ReembedReport rep;
    token_log_close();
    const std::string log = dir_ + "/line_tokens.log";
    std::map<std::string, search::TokenCsr> cache;
    uint64_t log_fp = 0;
    if (FILE *f = fopen(log.c_str(), "rb")) {
        char magic[8];
        uint64_t reserved = 0;
        // record sizes come from the file: bound every one of them by what is left of it before allocating
        uint64_t file_size = 0;
        { struct stat st; if (fstat(fileno(f), &st) == 0) file_size = (uint64_t)st.st_size; }
        if (fread(magic, 1, 8, f) == 8 && memcmp(magic, "SMTTOK01", 8) == 0 && fread(&log_fp, 8, 1, f) == 1 && fread(&reserved, 8, 1, f) == 1) {
            for (;;) {
                uint32_t head[4];
                uint64_t n_ids = 0;
                if (fread(head, 4, 4, f) != 4 || head[0] != TOK_TAG || fread(&n_ids, 8, 1, f) != 1) break;
                const uint64_t at = (uint64_t)ftello(f), left = file_size > at ? file_size - at : 0;
                const uint64_t n_lens = head[2] == TOK_TOMBSTONE ? 0 : head[2];
                if (head[1] > left || n_lens > (left - head[1]) / 4 || n_ids > (left - head[1] - n_lens * 4) / 4) break;   // torn / corrupt tail
                std::string path(head[1], '\0');
                if (head[1] && fread(&path[0], 1, head[1], f) != head[1]) break;
                if (head[2] == TOK_TOMBSTONE) { cache.erase(path); continue; }
                search::TokenCsr t;
                t.lens.resize(head[2]);
                t.ids.resize(n_ids);
                if (head[2] && fread(t.lens.data(), 4, head[2], f) != head[2]) break;   // torn tail: ignore it
                if (n_ids && fread(t.ids.data(), 4, n_ids, f) != n_ids) break;
                cache[path] = std::move(t);
            }
        }
        fclose(f);
    }
    if (log_fp != 0 && log_fp != model.tokenizer_fingerprint())
        throw Error("the cached tokens were produced by a different tokenizer than this model's; re-embed from the source files");
    // every live document needs a record that matches its extent
    std::vector<std::pair<uint64_t, std::string>> order;
    for (auto &kv : extents_) {
        auto it = cache.find(kv.first);
        uint64_t sum = 0;
        if (it != cache.end()) for (uint32_t l : it->second.lens) sum += l;
        if (it == cache.end() || it->second.lens.size() != kv.second.n_rows || sum != it->second.ids.size()) rep.missing.push_back(kv.first);
        order.emplace_back(kv.second.first_row, kv.first);
    }
    if (!rep.missing.empty()) return rep;   // nothing changed
    std::sort(order.begin(), order.end());
    smt_sharded_corpus *fresh = nullptr;
    check(smt_sharded_corpus_create(group_, SMT_DIM, &fresh), "reembed");
    // the new extents are collected aside and swapped in only once the fresh corpus is complete: if an embed call fails
    // half way, extents_ still describes the corpus that is still there
    std::map<std::string, Extent> fresh_extents;
    try {
        // batches of about 16384 lines, like encode_with_args' batch size
        std::vector<uint32_t> ids;
        std::vector<uint64_t> offsets(1, 0);
        auto flush = [&]() {
            if (offsets.size() > 1) model.embed_tokens_into(ids.data(), offsets.data(), offsets.size() - 1, fresh);
            ids.clear();
            offsets.assign(1, 0);
        };
        for (auto &o : order) {
            const search::TokenCsr &t = cache[o.second];
            fresh_extents[o.second] = Extent{smt_sharded_corpus_rows(fresh) + (offsets.size() - 1), extents_[o.second].n_rows};
            ids.insert(ids.end(), t.ids.begin(), t.ids.end());
            for (uint32_t l : t.lens) offsets.push_back(offsets.back() + l);
            rep.documents += 1;
            rep.lines += t.lens.size();
            rep.tokens += t.ids.size();
            if (offsets.size() > 16384) flush();
        }
        flush();
    } catch (...) { smt_sharded_corpus_destroy(fresh); throw; }
    drop_index();                            // (before the corpus it points into goes away)
    smt_sharded_corpus_destroy(corpus_);
    corpus_ = fresh;
    extents_.swap(fresh_extents);
    dead_rows_ = 0;
    rows_on_disk_valid_ = false;
    flush_line_embeddings();
    // rewrite the log with one record per live document (drops superseded records and tombstones)
    const std::string tmp = log + ".tmp";
    {
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f) throw Error("cannot open " + tmp + ": " + strerror(errno));
        const uint64_t fp = model.tokenizer_fingerprint(), zero = 0;
        bool ok = fwrite("SMTTOK01", 1, 8, f) == 8 && fwrite(&fp, 8, 1, f) == 1 && fwrite(&zero, 8, 1, f) == 1;
        for (auto &o : order) {
            const search::TokenCsr &t = cache[o.second];
            const uint32_t head[4] = {TOK_TAG, (uint32_t)o.second.size(), (uint32_t)t.lens.size(), 0};
            const uint64_t n_ids = t.ids.size();
            ok = ok && fwrite(head, 4, 4, f) == 4 && fwrite(&n_ids, 8, 1, f) == 1 && fwrite(o.second.data(), 1, o.second.size(), f) == o.second.size();
            if (!t.lens.empty()) ok = ok && fwrite(t.lens.data(), 4, t.lens.size(), f) == t.lens.size();
            if (n_ids) ok = ok && fwrite(t.ids.data(), 4, n_ids, f) == n_ids;
        }
        ok = (fclose(f) == 0) && ok;
        if (!ok || rename(tmp.c_str(), log.c_str()) != 0) throw Error("cannot rewrite " + log);
        token_log_fingerprint_ = fp;
    }
    return rep;
}

void Store::compact_if_sparse()
{
    const uint64_t total = smt_sharded_corpus_rows(corpus_);
    if (dead_rows_ < 4096 || dead_rows_ * 2 < total) return;
    // rewrite live extents back to back (row order of surviving documents is preserved)
    std::vector<std::pair<uint64_t, std::string>> order;
    for (auto &kv : extents_) order.emplace_back(kv.second.first_row, kv.first);
    std::sort(order.begin(), order.end());
    smt_sharded_corpus *fresh = nullptr;
    check(smt_sharded_corpus_create(group_, SMT_DIM, &fresh), "compact");
    std::map<std::string, Extent> fresh_extents;
    try {
        // extents travel in runs of up to 64 Ki rows: one read / one dealt append per run, not per document
        std::vector<float> buf;
        std::vector<std::pair<const std::string *, uint64_t>> run;   // (path, offset inside the run)
        uint64_t run_rows = 0;
        auto flush = [&]() {
            if (!run_rows) return;
            uint64_t first = 0;
            check(smt_sharded_corpus_append_host(fresh, buf.data(), run_rows, &first), "compact");
            for (auto &r : run) fresh_extents[*r.first] = Extent{first + r.second, extents_[*r.first].n_rows};
            run.clear();
            run_rows = 0;
        };
        for (auto &o : order) {
            const Extent &x = extents_[o.second];
            if (run_rows && run_rows + x.n_rows > 65536) flush();
            buf.resize((size_t)(run_rows + x.n_rows) * LINE_EMBEDDING_SIZE);
            check(smt_sharded_corpus_read_rows(corpus_, x.first_row, x.n_rows, buf.data() + (size_t)run_rows * LINE_EMBEDDING_SIZE), "compact");
            run.emplace_back(&o.second, run_rows);
            run_rows += x.n_rows;
        }
        flush();
    } catch (...) { smt_sharded_corpus_destroy(fresh); throw; }
    drop_index();                            // the index names rows by position and points into corpus_: it goes first
    smt_sharded_corpus_destroy(corpus_);
    corpus_ = fresh;
    extents_.swap(fresh_extents);
    dead_rows_ = 0;
    rows_on_disk_valid_ = false;  // rows moved: the file must be rewritten
}

WorkspaceStats Store::get_stats() const
{
    WorkspaceStats st;
    st.total_documents = count_documents();
    // The reference prints a hard-coded "HNSW" (store.rs:437-445) although its store scans exactly; here the line
    // says what is there: the IVF index once a workspace is large enough to have one, "No" for the exact scan.
    st.has_index = has_index();
    if (st.has_index) st.index_type = "IVF_PQ";
    return st;
}

std::vector<std::string> Store::get_all_document_paths() const
{
    std::vector<std::string> out;
    for (auto &kv : docs_) out.push_back(kv.first);
    return out;
}

std::vector<RankedLine> Store::search_line_embeddings(const std::vector<float> &query_vec,
                                                      const std::vector<std::string> &subset_paths, size_t top_k,
                                                      std::optional<float> max_distance) const
{
    std::vector<RankedLine> out;
    if (subset_paths.empty() || top_k == 0) return out;  // store.rs:489-491
    if (query_vec.size() != LINE_EMBEDDING_SIZE) throw Error("query vector must have 256 dimensions");
    struct Seg { uint64_t first, n; const std::string *path; };
    std::vector<Seg> segs;
    for (auto &p : subset_paths) {
        auto it = extents_.find(p);
        if (it != extents_.end() && it->second.n_rows) segs.push_back({it->second.first_row, it->second.n_rows, &it->first});
    }
    if (segs.empty()) return out;
    std::sort(segs.begin(), segs.end(), [](const Seg &a, const Seg &b) { return a.first < b.first; });
    segs.erase(std::unique(segs.begin(), segs.end(), [](const Seg &a, const Seg &b) { return a.first == b.first; }), segs.end());
    std::vector<smt_range> ranges;
    for (auto &s : segs) {
        if (!ranges.empty() && ranges.back().end == s.first) ranges.back().end = s.first + s.n;
        else ranges.push_back({s.first, s.first + s.n});
    }
    const uint32_t k = (uint32_t)std::min<size_t>(top_k, 0xFFFFFFFFu);
    std::vector<uint64_t> rows(top_k);
    std::vector<double> dist(top_k);
    uint64_t n = 0;
    bool answered = false;
    // ---- approximate path: whole-workspace search over a large store (see set_index_policy)
    uint64_t ranged = 0;
    for (auto &r : ranges) ranged += r.end - r.begin;
    if (smt_sharded_corpus_rows(corpus_) >= index_min_rows_ && top_k <= 24 && ranged == count_line_embeddings() && ensure_index()) {
        const uint32_t fetch = (uint32_t)std::min<size_t>(56, 2 * top_k + 8);  // head-room for dead rows and the threshold
        const uint32_t rerank = (uint32_t)std::min<size_t>(512, std::max<size_t>(64, 2 * top_k * oversample_factor_));
        std::vector<uint64_t> c_rows(fetch);
        std::vector<double> c_dist(fetch);
        uint64_t c_n = 0;
        uint32_t n_lists = 0;
        check(smt_sharded_ivfpq_info(index_, nullptr, &n_lists, nullptr), "search_line_embeddings (index)");
        check(smt_sharded_ivfpq_search(index_, query_vec.data(), 1, fetch, std::min<uint32_t>(std::min<uint32_t>(index_nprobe_, n_lists), 512),
                                       rerank, c_rows.data(), c_dist.data(), &c_n, fetch), "search_line_embeddings (index)");
        const float thr_score = max_distance ? 1.0f - *max_distance : 0.0f;
        bool cut_by_threshold = false;
        for (uint64_t i = 0; i < c_n && n < top_k; ++i) {
            auto it = std::upper_bound(segs.begin(), segs.end(), c_rows[i], [](uint64_t r, const Seg &s) { return r < s.first; });
            if (it == segs.begin() || c_rows[i] >= (it - 1)->first + (it - 1)->n) continue;       // a dead row (replaced document)
            if (max_distance && !((1.0 - c_dist[i]) > (double)thr_score)) { cut_by_threshold = true; break; }   // store.rs:502-503 (sorted: the rest fails too)
            rows[n] = c_rows[i];
            dist[n] = c_dist[i];
            ++n;
        }
        // A short list is only an answer when the THRESHOLD cut it (the candidates are sorted by exact distance, so
        // nothing behind the cut passes either).  Otherwise -- dead rows crowded the list, the probed lists held fewer
        // than top_k rows -- the exact scan answers.
        answered = n == top_k || cut_by_threshold;
        if (!answered) n = 0;
    }
    if (!answered)
        check(smt_sharded_search(corpus_, query_vec.data(), 1, k, max_distance ? (double)*max_distance : NAN, SMT_MODE_WORKSPACE,
                                 ranges.data(), (uint32_t)ranges.size(), rows.data(), dist.data(), &n, top_k),
              "search_line_embeddings");
    for (uint64_t i = 0; i < n; ++i) {
        auto it = std::upper_bound(segs.begin(), segs.end(), rows[i], [](uint64_t r, const Seg &s) { return r < s.first; });
        const Seg &s = *(it - 1);
        RankedLine rl;
        rl.path = *s.path;
        rl.line_number = (int32_t)(rows[i] - s.first);
        rl.distance = (float)dist[i];  // 1 - score as f32 (store.rs:531)
        out.push_back(std::move(rl));
    }
    return out;
}

std::vector<DocumentState> Store::analyze_document_states(const std::vector<std::string> &file_paths) const
{
    const auto existing = get_existing_docs(file_paths);
    std::vector<DocumentState> states;
    for (auto &fp : file_paths) {
        struct stat st;
        if (stat(fp.c_str(), &st) != 0) continue;  // file doesn't exist: skipped (store.rs:573-576)
        DocMeta cur;
        cur.path = fp;
        cur.size_bytes = (uint64_t)st.st_size;
        cur.mtime = (int64_t)st.st_mtime;
        cur._version = CURRENT_EMBEDDING_VERSION;
        auto it = existing.find(fp);
        DocumentState ds;
        if (it != existing.end()) {
            const DocMeta &ex = it->second;
            // (a document whose line rows are gone -- torn write, unreadable corpus file -- must be re-embedded even
            // though its metadata says "unchanged": it would otherwise silently drop out of every search)
            const bool rows_missing = extents_.find(fp) == extents_.end();
            if (rows_missing || ex.size_bytes != cur.size_bytes || ex.mtime != cur.mtime || ex._version != CURRENT_EMBEDDING_VERSION) {
                ds.kind = DocumentState::Changed;
                ds.info = DocumentInfo{fp, read_to_string(fp), cur};
            } else {
                ds.kind = DocumentState::Unchanged;
                ds.filename = fp;
            }
        } else {
            ds.kind = DocumentState::New;
            ds.info = DocumentInfo{fp, read_to_string(fp), cur};
        }
        states.push_back(std::move(ds));
    }
    return states;
}

size_t Store::count_documents() const { return docs_.size(); }

size_t Store::count_line_embeddings() const
{
    size_t n = 0;
    for (auto &kv : extents_) n += (size_t)kv.second.n_rows;
    return n;
}

void Store::flush_documents() const
{
    json::Value arr = json::Value::array();
    for (auto &kv : docs_) {
        json::Value d = json::Value::object();
        d.set("path", json::Value::str(kv.second.path));
        d.set("size_bytes", json::Value::uint(kv.second.size_bytes));
        d.set("mtime", json::Value::sint(kv.second.mtime));
        d.set("_version", json::Value::uint(kv.second._version));
        arr.arr.push_back(std::move(d));
    }
    write_file_atomic(dir_ + "/documents.json", json::to_string_pretty(arr));
}

void Store::flush_line_embeddings() const
{
    token_log_close();
    // vectors first, then the extent table that references them (a crash in between leaves extra
    // rows that no extent points at -- harmless; the reverse order could reference missing rows)
    const std::string emb = dir_ + "/line_embeddings.f32";
    const uint64_t rows = smt_sharded_corpus_rows(corpus_);
    if (rows_on_disk_valid_ && rows >= rows_on_disk_ && path_exists(emb)) {
        if (rows > rows_on_disk_) check(smt_sharded_corpus_append_to_file(corpus_, emb.c_str(), rows_on_disk_), "flush_line_embeddings");
    } else {
        check(smt_sharded_corpus_save(corpus_, emb.c_str()), "flush_line_embeddings");  // first flush or after a compaction
    }
    rows_on_disk_ = rows;
    rows_on_disk_valid_ = true;
    json::Value root = json::Value::object();
    json::Value arr = json::Value::array();
    for (auto &kv : extents_) {
        json::Value e = json::Value::object();
        e.set("path", json::Value::str(kv.first));
        e.set("first_row", json::Value::uint(kv.second.first_row));
        e.set("n_rows", json::Value::uint(kv.second.n_rows));
        arr.arr.push_back(std::move(e));
    }
    root.set("extents", std::move(arr));
    int n_ranks = 1;
    (void)smt_group_info(group_, &n_ranks, nullptr, nullptr, nullptr, nullptr);
    if (n_ranks > 1) {   // how the rows are dealt over the GPUs: [rows, rank] per piece, in global row order (see Store::open)
        const uint64_t n_pieces = smt_sharded_corpus_layout(corpus_, nullptr, nullptr, 0);
        std::vector<uint64_t> piece_rows(n_pieces);
        std::vector<uint32_t> piece_rank(n_pieces);
        (void)smt_sharded_corpus_layout(corpus_, piece_rows.data(), piece_rank.data(), n_pieces);
        json::Value pieces = json::Value::array();
        for (uint64_t k = 0; k < n_pieces; ++k) {
            json::Value pc = json::Value::array();
            pc.arr.push_back(json::Value::uint(piece_rows[k]));
            pc.arr.push_back(json::Value::uint(piece_rank[k]));
            pieces.arr.push_back(std::move(pc));
        }
        json::Value sh = json::Value::object();
        sh.set("n_ranks", json::Value::uint((uint64_t)n_ranks));
        sh.set("pieces", std::move(pieces));
        root.set("shards", std::move(sh));
    }
    write_file_atomic(dir_ + "/line_rows.json", json::to_string_pretty(root));
}

}  // namespace workspace

// ================================================================== output
namespace cmds {

static void push_line(std::string &out, size_t line_number_1based, const std::string &line, bool highlight)
{
    char num[32];
    snprintf(num, sizeof(num), "%4zu: ", line_number_1based);  // "{:4}: {}"
    if (highlight) out += "\x1b[43m\x1b[30m";
    out += num;
    out += line;
    if (highlight) out += "\x1b[0m";
    out += "\n";
}

std::string print_search_results(const std::vector<search::SearchResult> &results, bool is_tty)
{
    std::string out;
    for (auto &r : results) {
        out += r.filename + ":" + std::to_string(r.start) + "::" + std::to_string(r.end) + " (" +
               fmt::rust_display(r.distance) + ")\n";                                   // search.rs:43
        for (size_t i = 0; i < r.lines.size(); ++i) {
            const size_t line_number = r.start + i;
            push_line(out, line_number + 1, r.lines[i], is_tty && line_number == r.match_line);  // :47-59
        }
        out += "\n";                                                                     // :61
    }
    return out;
}

std::string print_workspace_search_results(const std::vector<workspace::RankedLine> &ranked, size_t n_lines, bool is_tty)
{
    std::string out;
    for (auto &rl : ranked) {
        const size_t match = (size_t)rl.line_number;
        const size_t start = match > n_lines ? match - n_lines : 0;
        const size_t end = match + n_lines + 1;  // NOT clamped in the header (search.rs:77-79)
        out += rl.path + ":" + std::to_string(start) + "::" + std::to_string(end) + " (" + fmt::rust_display(rl.distance) + ")\n";
        std::string content;
        bool ok = true;
        try { content = read_to_string(rl.path); } catch (const Error &) { ok = false; }
        if (ok) {
            const std::vector<std::string> lines = lines_of(content);
            const size_t actual_end = std::min(end, lines.size());
            // the reference slices lines[start..actual_end] and panics if start > len (stale rows);
            // we print nothing for that window instead ("do not replicate")
            for (size_t ln = start; ln < actual_end; ++ln) push_line(out, ln + 1, lines[ln], is_tty && ln == match);
        } else {
            out += "    [Error: Could not read file content]\n";
        }
        out += "\n";
    }
    return out;
}

static json::Value result_json(const std::string &filename, size_t start, size_t end, size_t match, double distance,
                               const std::string &content)
{
    json::Value o = json::Value::object();  // field order: src/json_mode.rs:17-30
    o.set("filename", json::Value::str(filename));
    o.set("start_line_number", json::Value::uint(start));
    o.set("end_line_number", json::Value::uint(end));
    o.set("match_line_number", json::Value::uint(match));
    o.set("distance", json::Value::num(distance));
    o.set("content", json::Value::str(content));
    return o;
}

static std::string join_lines(const std::vector<std::string> &lines, size_t b, size_t e)
{
    std::string s;
    for (size_t i = b; i < e; ++i) { if (i > b) s += "\n"; s += lines[i]; }
    return s;
}

std::string search_results_json(const std::vector<search::SearchResult> &results)
{
    json::Value arr = json::Value::array();
    for (auto &r : results)
        arr.arr.push_back(result_json(r.filename, r.start, r.end, r.match_line, r.distance, join_lines(r.lines, 0, r.lines.size())));
    json::Value root = json::Value::object();
    root.set("results", std::move(arr));
    return json::to_string_pretty(root) + "\n";
}

std::string workspace_results_json(const std::vector<workspace::RankedLine> &ranked, size_t n_lines)
{
    json::Value arr = json::Value::array();
    for (auto &rl : ranked) {
        const size_t match = (size_t)rl.line_number;
        const size_t start = match > n_lines ? match - n_lines : 0;
        const size_t end = match + n_lines + 1;
        std::string content;
        try {
            const std::vector<std::string> lines = lines_of(read_to_string(rl.path));
            const size_t actual_end = std::min(end, lines.size());
            content = start <= actual_end ? join_lines(lines, start, actual_end) : "";
        } catch (const Error &) {
            content = "[Error: Could not read file content]";
        }
        arr.arr.push_back(result_json(rl.path, start, end, match, (double)rl.distance, content));  // `as f64` (search.rs:233)
    }
    json::Value root = json::Value::object();
    root.set("results", std::move(arr));
    return json::to_string_pretty(root) + "\n";
}

}  // namespace cmds
}  // namespace semtools
