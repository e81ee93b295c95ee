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
#include <condition_variable>
#include <cstring>
#include <deque>
#include <exception>
#include <fstream>
#include <mutex>
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
std::vector<std::string_view> line_views(std::string_view content)
{
    // str::lines() (mod.rs:51): split at '\n', a '\r' before it belongs to the line ending, a last line without terminator is kept.
    // A line ends where a '\n' is, whatever came before: big contents are cut into byte slices whose '\n' positions are found on several
    // threads (memchr over 64 MB was 14 ms of a 55 ms call on one), then the views are laid out from the per-slice counts.
    const size_t n = content.size();
    const char *base = content.data();
    std::vector<std::string_view> out;
    if (n == 0) return out;
    const size_t n_slices = n >= (4u << 20) ? std::min<size_t>(8, std::max<size_t>(1, std::thread::hardware_concurrency())) : 1;
    std::vector<std::vector<size_t>> nl(n_slices);   // positions of the '\n's of each slice
    auto find = [&](size_t t) {
        const size_t b = n * t / n_slices, e = n * (t + 1) / n_slices;
        std::vector<size_t> v;   // (grown in this thread's frame, handed over once: the headers of nl[] share cache lines)
        v.reserve((e - b) / 48 + 16);
        for (size_t at = b; at < e;) {
            const void *p = memchr(base + at, '\n', e - at);
            if (!p) break;
            const size_t pos = (size_t)(static_cast<const char *>(p) - base);
            v.push_back(pos);
            at = pos + 1;
        }
        nl[t] = std::move(v);
    };
    if (n_slices == 1) find(0);
    else parallel_slices(n_slices, 1, [&](size_t b, size_t e) { for (size_t t = b; t < e; ++t) find(t); });
    std::vector<size_t> first(n_slices + 1, 0);      // index of each slice's first line end
    for (size_t t = 0; t < n_slices; ++t) first[t + 1] = first[t] + nl[t].size();
    const size_t terminated = first[n_slices];
    const bool tail = base[n - 1] != '\n';           // a last line without terminator
    out.resize(terminated + (tail ? 1 : 0));
    auto lay = [&](size_t t) {
        // the line that ends at nl[t][j] starts right after the previous '\n' (the last one of an earlier non-empty slice, or 0)
        size_t start = 0;
        for (size_t u = t; u-- > 0;)
            if (!nl[u].empty()) { start = nl[u].back() + 1; break; }
        for (size_t j = 0; j < nl[t].size(); ++j) {
            size_t end = nl[t][j];
            const size_t next = end + 1;
            if (end > start && base[end - 1] == '\r') --end;   // "\r\n"
            out[first[t] + j] = std::string_view(base + start, end - start);
            start = next;
        }
    };
    if (n_slices == 1) lay(0);
    else parallel_slices(n_slices, 1, [&](size_t b, size_t e) { for (size_t t = b; t < e; ++t) lay(t); });
    if (tail) {
        size_t start = 0;
        for (size_t u = n_slices; u-- > 0;)
            if (!nl[u].empty()) { start = nl[u].back() + 1; break; }
        out[terminated] = std::string_view(base + start, n - start);   // (a lone '\r' at the very end stays: no '\n' follows it)
    }
    return out;
}

std::vector<std::string> lines_of(std::string_view content)
{
    // the strings are built on several threads for big files (a million small allocations were three quarters of the wall time of
    // embedding a 1 M-line file before round 4; the search path itself keeps views and builds none)
    const std::vector<std::string_view> views = line_views(content);
    std::vector<std::string> out(views.size());
    auto build = [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) out[i].assign(views[i].data(), views[i].size());
    };
    if (views.size() >= 65536) parallel_slices(views.size(), 16384, build);
    else build(0, views.size());
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

// fn(offset, length) for every maximal run of non-whitespace bytes (isspace of the "C" locale, whatever locale the host process has
// set: ' ', \t, \n, \v, \f, \r).  No span list: a vector per line was a heap allocation and five regrowths per line -- most of the
// 1.4 us a thread spent per line of a big file (profiles/r04_ingest_phases.json).
inline bool is_space_c(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
template <typename Fn>
inline void for_each_word(const std::string &text, Fn fn)
{
    size_t i = 0;
    const size_t n = text.size();
    const char *p = text.data();
    while (i < n) {
        while (i < n && is_space_c((unsigned char)p[i])) ++i;
        const size_t s = i;
        while (i < n && !is_space_c((unsigned char)p[i])) ++i;
        if (i > s) fn(s, i - s);
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
        for_each_word(text, [&](size_t at, size_t len) {
            const int64_t hit = vocab_.find(text.data() + at, len);
            if (hit >= 0) ids.push_back((uint32_t)hit);
            else if (unk_) ids.push_back(*unk_);
        });
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
        for_each_word(text, [&](size_t at, size_t len) {
            ids.push_back((uint32_t)(smt_fnv1a_hash(reinterpret_cast<const uint8_t *>(text.data() + at), len) % v_));
        });
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
    try {
        parallel_slices(uniq.size(), 512, [&](size_t sb, size_t se) {
            for (size_t s = sb; s < se; ++s) {
                const off_t at = (off_t)(lazy_offset_ + (uint64_t)uniq[s] * SMT_DIM * sizeof(float));
                if (pread(lazy_fd_, &compact[s * SMT_DIM], SMT_DIM * sizeof(float), at) != (ssize_t)(SMT_DIM * sizeof(float)))
                    throw Error("short read from " + lazy_path_);
            }
        });
    } catch (...) {
        for (uint32_t id : uniq) lazy_slot_[id] = 0;
        throw;
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
std::mutex g_phase_mu;   // (add() is called from the embedding pipeline's tokenizer thread)
}  // namespace
void PhaseTimer::add(const char *phase, double ms)
{
    std::lock_guard<std::mutex> lk(g_phase_mu);
    for (auto &p : g_phases)
        if (p.first == phase) { p.second += ms; return; }
    g_phases.emplace_back(phase, ms);
}
void PhaseTimer::mark(const char *phase)
{
    const auto now = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(now - g_phase_t0).count();
    g_phase_t0 = now;
    add(phase, ms);
}
std::string PhaseTimer::json()
{
    std::lock_guard<std::mutex> lk(g_phase_mu);
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
static size_t truncate_len(std::string_view s, size_t max_tokens, size_t median_len)
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

void StaticModel::tokenize_batch(const std::string_view *sentences, size_t begin, size_t end,
                                 std::optional<size_t> max_length, std::vector<uint32_t> &ids,
                                 std::vector<uint64_t> &offsets) const
{
    // encode_batch_fast is rayon-parallel upstream; here: one contiguous slice of the batch per thread, in two steps with a barrier
    // between them -- (A) tokenise the slice into the thread's own buffers, [the last thread to arrive sizes the batch's arrays from
    // the slices' token counts], (B) copy the slice's ids to their place and write its offsets.  (Round 4: 128 threads of 2048 lines
    // and a serial merge of the parts took 11 ms per 262144-line batch of which 0.3 ms was tokenising -- starting a thread costs
    // ~30 us, a whitespace-hashed line ~0.15 us: profiles/r04_ingest_phases.json.)
    const size_t n = end - begin;
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), (size_t)64, n / std::max<size_t>(tok_->lines_per_thread(), 1)}));
    std::vector<std::vector<uint32_t>> part_ids(n_threads);
    std::vector<std::vector<uint32_t>> part_len(n_threads);
    std::vector<uint64_t> id_base(n_threads + 1, 0);
    const auto unk = tok_->unk_id();
    // a throwing tokenizer (callback failure, bad_alloc) must reach the caller -- and through it the extern "C"
    // wrappers' catch blocks -- not std::terminate the process from a worker thread (and it must still arrive at the barrier)
    std::vector<std::exception_ptr> failed(n_threads);
    std::mutex mu;
    std::condition_variable cv;
    size_t arrived = 0;
    bool sized = false, any_failed = false;
    auto slice = [&](size_t t) { return std::make_pair(begin + n * t / n_threads, begin + n * (t + 1) / n_threads); };
    auto tokenise = [&](size_t t) {   // (A)
        const auto [b, e] = slice(t);
        try {
            std::vector<uint32_t> tmp;
            std::string text;   // (the tokenizers take a std::string: one buffer per thread, refilled per line -- no allocation once it has grown)
            // the slice's ids and lengths grow in vectors of THIS thread's frame and are handed over once: the headers of
            // part_ids[t] / part_len[t] of neighbouring threads share cache lines, and an append per line into them made every
            // thread's line cost 0.7 us instead of 0.25 (false sharing; profiles/r04_ingest_phases.json)
            std::vector<uint32_t> my_ids, my_len;
            my_len.reserve(e - b);
            my_ids.reserve((e - b) * 16);
            const size_t median = tok_->median_token_length();
            for (size_t i = b; i < e; ++i) {
                tmp.clear();
                const std::string_view src = sentences[i];
                const size_t keep = max_length ? truncate_len(src, *max_length, median) : src.size();
                text.assign(src.data(), keep);
                tok_->encode(text, tmp);
                if (unk) tmp.erase(std::remove(tmp.begin(), tmp.end(), *unk), tmp.end());
                if (max_length && tmp.size() > *max_length) tmp.resize(*max_length);
                my_ids.insert(my_ids.end(), tmp.begin(), tmp.end());
                my_len.push_back((uint32_t)tmp.size());
            }
            part_ids[t] = std::move(my_ids);
            part_len[t] = std::move(my_len);
        } catch (...) { failed[t] = std::current_exception(); }
    };
    // the barrier: slices [t0, t0 + count) of the calling thread are done; the last thread to arrive lays the batch out.  Returns
    // whether (B) may run.
    auto arrive = [&](size_t t0, size_t count) -> bool {
        std::unique_lock<std::mutex> lk(mu);
        for (size_t t = t0; t < t0 + count; ++t) if (failed[t]) any_failed = true;
        arrived += count;
        if (arrived == n_threads) {
            if (!any_failed) {
                try {
                    for (size_t u = 0; u < n_threads; ++u) id_base[u + 1] = id_base[u] + part_ids[u].size();
                    ids.resize(id_base[n_threads]);
                    offsets.resize(n + 1);
                    offsets[0] = 0;
                } catch (...) { failed[t0] = std::current_exception(); any_failed = true; }
            }
            sized = true;
            cv.notify_all();
        } else cv.wait(lk, [&] { return sized; });
        return !any_failed;
    };
    auto place = [&](size_t t) {      // (B)
        if (!part_ids[t].empty()) memcpy(ids.data() + id_base[t], part_ids[t].data(), part_ids[t].size() * sizeof(uint32_t));
        uint64_t at = id_base[t];
        uint64_t *off = offsets.data() + (slice(t).first - begin) + 1;
        for (uint32_t l : part_len[t]) { at += l; *off++ = at; }
    };
    auto work = [&](size_t t) {
        tokenise(t);
        if (arrive(t, 1)) place(t);
    };
    // the LAST slice runs on the calling thread; a thread that cannot be started leaves its slice -- and all later ones -- to the
    // caller too, who arrives for all of them at once: nobody waits for a thread that does not exist
    std::vector<std::thread> th;
    size_t started = 0;
    try {
        for (; started + 1 < n_threads; ++started) th.emplace_back(work, started);
    } catch (const std::system_error &) {}
    for (size_t t = started; t < n_threads; ++t) tokenise(t);
    if (arrive(started, n_threads - started))
        for (size_t t = started; t < n_threads; ++t) place(t);
    for (auto &x : th) x.join();
    for (auto &f : failed) if (f) std::rethrow_exception(f);
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
    const std::vector<std::string_view> views(sentences.begin(), sentences.end());
    for (size_t b = 0; b < sentences.size(); b += batch_size) {
        const size_t e = std::min(sentences.size(), b + batch_size);
        tokenize_batch(views.data(), b, e, max_length, ids, offsets);
        buf.resize((e - b) * SMT_DIM);
        embed_csr(ids, offsets, e - b, buf.data(), nullptr);   // (tokenize_batch already truncated to max_length)
        for (size_t i = 0; i < e - b; ++i) out.emplace_back(buf.begin() + i * SMT_DIM, buf.begin() + (i + 1) * SMT_DIM);
    }
    return out;
}

uint64_t StaticModel::encode_into(const std::vector<std::string> &sentences, std::optional<size_t> max_length,
                                  size_t batch_size, smt_sharded_corpus *corpus, TokenCsr *sink, const std::function<void()> *after_batch) const
{
    return encode_into(std::vector<std::string_view>(sentences.begin(), sentences.end()), max_length, batch_size, corpus, sink, after_batch);
}

uint64_t StaticModel::encode_into(const std::vector<std::string_view> &sentences, std::optional<size_t> max_length,
                                  size_t batch_size, smt_sharded_corpus *corpus, TokenCsr *sink, const std::function<void()> *after_batch) const
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
    // ("within_*": the two sides of the pipeline, overlapping each other inside the caller's tokenize_and_embed phase)
    auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    auto tokenize = [&](size_t b, Slot &s) {
        const auto t0 = std::chrono::steady_clock::now();
        tokenize_batch(sentences.data(), b, std::min(n, b + batch_size), max_length, s.ids, s.offsets);
        PhaseTimer::add("within_embed:tokenize_batches", ms_since(t0));
    };
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
        const auto t_embed = std::chrono::steady_clock::now();
        try { embed_csr(slots[cur].ids, slots[cur].offsets, e - b, nullptr, corpus); } catch (...) { embed_failed = std::current_exception(); }
        PhaseTimer::add("within_embed:upload_and_K1", ms_since(t_embed));
        if (!embed_failed && after_batch && *after_batch) {   // (the tokenizer threads of the next batch are running meanwhile)
            const auto t_after = std::chrono::steady_clock::now();
            try { (*after_batch)(); } catch (...) { embed_failed = std::current_exception(); }
            PhaseTimer::add("within_embed:after_batch", ms_since(t_after));
        }
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
    std::vector<std::string_view> lines = line_views(content);
    if (lines.empty()) return std::nullopt;  // mod.rs:57-59
    PhaseTimer::mark("split_lines");
    Document doc;
    doc.filename = filename;
    if (ignore_case) {
        std::vector<std::string> lowered(lines.size());
        parallel_slices(lines.size(), 16384, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) lowered[i] = to_lowercase(std::string(lines[i]));
        });
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

std::vector<Document> load_documents(const std::vector<std::string> &files, const StaticModel &model, bool ignore_case, Embeddings &emb)
{
    // mod.rs:128-134 reads and embeds file by file; the rows do not depend on how the lines are batched, so all files
    // are read first (the first error still aborts before anything is printed) and embedded in ONE pipeline run
    // (tokenise || H2D || K1 across file borders instead of a round of threads and a K1 launch per file)
    std::vector<Document> documents;
    std::vector<std::string_view> all;
    std::deque<std::string> lowered;   // (mod.rs:61-67: the lowered copies are what is embedded; a deque never moves its elements)
    // the files are read (and cut into lines) on up to eight threads; the first unreadable file IN ORDER is the error (mod.rs:130 `?`)
    std::vector<std::shared_ptr<const std::string>> texts(files.size());
    std::vector<std::vector<std::string_view>> views(files.size());
    std::vector<std::exception_ptr> unreadable(files.size());
    parallel_slices(files.size(), 8, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            try {
                texts[i] = std::make_shared<const std::string>(read_to_string(files[i]));
                views[i] = line_views(*texts[i]);
            } catch (...) { unreadable[i] = std::current_exception(); }
        }
    });
    for (auto &u : unreadable) if (u) std::rethrow_exception(u);
    for (size_t fi = 0; fi < files.size(); ++fi) {
        const std::string &f = files[fi];
        std::shared_ptr<const std::string> text = std::move(texts[fi]);
        std::vector<std::string_view> lines = std::move(views[fi]);
        if (lines.empty()) continue;                     // create_document_from_content -> None (mod.rs:57-59)
        Document doc;
        doc.filename = f;
        doc.first_row = all.size();
        if (ignore_case) {
            for (auto &l : lines) {
                lowered.push_back(to_lowercase(std::string(l)));
                all.push_back(lowered.back());
            }
        } else all.insert(all.end(), lines.begin(), lines.end());
        doc.lines = std::move(lines);
        doc.text = std::move(text);
        documents.push_back(std::move(doc));
    }
    PhaseTimer::mark("split_lines");
    if (!all.empty()) {
        const uint64_t first = model.encode_into(all, 2048, 16384, emb.corpus());     // mod.rs:69
        for (auto &d : documents) d.first_row += first;
    }
    PhaseTimer::mark("tokenize_and_embed");
    return documents;
}

std::vector<SearchResult> search_files(const std::vector<std::string> &files, const std::string &query,
                                       const StaticModel &model, const SearchConfig &config)
{
    PhaseTimer::mark("model_load");
    Embeddings emb(model.group());
    const std::vector<Document> documents = load_documents(files, model, config.ignore_case, emb);
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
    std::vector<std::pair<std::string, std::vector<std::string_view>>> pending;   // views into doc_states' contents ...
    std::deque<std::string> lowered;                                                // ... or into the lowered copies (mod.rs:61-67)
    for (auto &st : doc_states) {
        if (st.kind == DocumentState::Unchanged) continue;
        std::vector<std::string_view> lines = line_views(st.info.content);
        if (lines.empty()) continue;  // create_document_from_content -> None
        if (config.ignore_case)
            for (auto &s : lines) {
                lowered.push_back(to_lowercase(std::string(s)));
                s = lowered.back();
            }
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
