// host.cpp -- C++ mirror of the reference's search module (src/search/mod.rs) on top of the C ABI (see host.h): strings,
// tokenizers, StaticModel, Documents, search_documents / search_files / search_with_workspace.  The workspace and its
// store live in store.cpp, the output formats in output.cpp.  String handling, file I/O and bookkeeping only: every
// floating-point result comes from libsemtools_hip's kernels.
#include "host.h"
#include "host_internal.h"
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

void check(int rc, const char *what)
{
    if (rc != SMT_OK) throw Error(std::string(what) + ": " + smt_last_error());
}

// ------------------------------------------------------------------ strings / files
std::vector<std::string> lines_of(std::string_view content)
{
    // str::lines() (mod.rs:51): split at '\n', a '\r' before it belongs to the line ending, a last line without terminator is kept.
    // Pass 1 finds the line starts; pass 2 builds the strings -- on several threads for big files (a million small allocations were
    // three quarters of the wall time of embedding a 1 M-line file: profiles/r04_ingest.json).
    std::vector<size_t> starts;
    const size_t n = content.size();
    for (size_t start = 0; start < n;) {
        starts.push_back(start);
        const void *nl = memchr(content.data() + start, '\n', n - start);
        if (!nl) break;
        start = (size_t)(static_cast<const char *>(nl) - content.data()) + 1;
    }
    std::vector<std::string> out(starts.size());
    auto build = [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const size_t start = starts[i];
            size_t end = i + 1 < starts.size() ? starts[i + 1] - 1 : n;       // the '\n' (or the end of the content)
            if (i + 1 == starts.size() && end > start && content[end - 1] == '\n') --end;   // (last line WITH a terminator)
            if (end > start && content[end - 1] == '\r' && (i + 1 < starts.size() || (end < n && content[end] == '\n'))) --end;  // "\r\n"
            out[i].assign(content.data() + start, end - start);
        }
    };
    if (starts.size() >= 65536) parallel_slices(starts.size(), 16384, build);
    else build(0, starts.size());
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

void write_file_atomic(const std::string &path, const std::string &data)
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

bool path_exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

void mkdir_p(const std::string &dir)
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
// (returns the number of BYTES to keep: almost every line is kept whole, and copying it just to hand it to the tokenizer was a heap
// allocation per line)
static size_t truncate_len(const std::string &s, size_t max_tokens, size_t median_len)
{
    const size_t max_chars = max_tokens * median_len;
    if (s.size() <= max_chars) return s.size();   // (at least one byte per character)
    size_t chars = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        if (((unsigned char)s[i] & 0xC0) != 0x80) {
            if (chars == max_chars) return i;
            ++chars;
        }
    }
    return s.size();
}

void StaticModel::tokenize_batch(const std::vector<std::string> &sentences, size_t begin, size_t end,
                                 std::optional<size_t> max_length, std::vector<uint32_t> &ids,
                                 std::vector<uint64_t> &offsets) const
{
    // encode_batch_fast is rayon-parallel upstream; here: one slice per hardware thread
    const size_t n = end - begin;
    // (at least 2048 lines per thread: starting a thread costs ~25 us, a whitespace-hashed line ~0.1 us -- 256 threads for a
    // 65536-line batch spent more time being started than tokenising: profiles/r04_ingest.json)
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), n / 2048));
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
            const std::string &src = sentences[i];
            const size_t keep = max_length ? truncate_len(src, *max_length, tok_->median_token_length()) : src.size();
            if (keep == src.size()) tok_->encode(src, tmp);
            else tok_->encode(src.substr(0, keep), tmp);
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
    // starting them: SEMTOOLS_EMBED_BATCH overrides (default: the caller's value, at least 262144).
    {
        static const size_t env_batch = [] { const char *e = getenv("SEMTOOLS_EMBED_BATCH"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)0; }();
        batch_size = env_batch ? env_batch : std::max<size_t>(batch_size, 262144);
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

std::optional<Document> create_document_from_content(const std::string &filename, std::string_view content,
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
    PhaseTimer::mark("session_device_search");
    parallel_slices(nq, 128, [&](size_t q_begin, size_t q_end) {   // (results of different queries share nothing)
    for (size_t q = q_begin; q < q_end; ++q) {
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
    });
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

}  // namespace semtools
