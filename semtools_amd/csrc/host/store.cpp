// store.cpp -- the reference's workspace (src/workspace/mod.rs) and its store (src/workspace/store.rs: the two Qdrant
// shards became documents.json + line_rows.json + line_embeddings.f32 [+ line_index.ivf*] + line_tokens.log) on an
// smt_sharded_corpus: host bookkeeping only, every vector lives on the GPUs of the caller's group.
#include "host.h"
#include "host_internal.h"

#include <dirent.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fmt.h"
#include "json.h"

namespace semtools {

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
    if (have_rows_json)
        if (auto *gen = rows_json.get("generation")) s->generation_ = gen->as_u64();
    // The index files name LOCAL rows by position: only valid on the layout they were built on AND for the rows as they lay when it
    // was saved.  `line_index.gen` (written after every index save) names the corpus generation -- bumped whenever rows move or are
    // rewritten in place -- and the rank count: an index left behind by a session with another number of GPUs, or one whose
    // removal failed, is never loaded (such an index would pass smt_ivfpq_load's range checks and lose recall silently).
    bool index_current = false;
    if (corpus_ok && path_exists(workspace_dir + "/line_index.gen")) {
        try {
            const json::Value g = json::parse(read_to_string(workspace_dir + "/line_index.gen"));
            auto *gg = g.get("generation");
            auto *gr = g.get("n_ranks");
            index_current = gg && gr && gg->as_u64() == s->generation_ && gr->as_u64() == (uint64_t)n_ranks;
        } catch (const std::exception &) {}
    }
    s->index_on_disk_ = corpus_ok && index_current && (n_ranks == 1 || layout_restored) && path_exists(s->index_file(0));
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
    // rows are about to move or change: a new corpus generation (flushed with line_rows.json), and EVERY index file goes -- also the
    // ones a session with a different number of GPUs wrote (`line_index.ivf` vs `line_index.ivf.r<r>of<n>`)
    ++generation_;
    remove_index_files();
    index_on_disk_ = false;
}

void Store::remove_index_files() const
{
    DIR *d = opendir(dir_.c_str());
    if (!d) return;
    std::vector<std::string> victims;
    while (const dirent *e = readdir(d))
        if (strncmp(e->d_name, "line_index.", 11) == 0) victims.push_back(dir_ + "/" + e->d_name);
    closedir(d);
    for (auto &v : victims) (void)remove(v.c_str());
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
        if (smt_sharded_ivfpq_save(index_, file.c_str()) == SMT_OK) {
            json::Value g = json::Value::object();
            g.set("generation", json::Value::uint(generation_));
            g.set("n_ranks", json::Value::uint((uint64_t)n_ranks));
            g.set("rows", json::Value::uint(rows));
            try {
                write_file_atomic(dir_ + "/line_index.gen", json::to_string_pretty(g));
                index_on_disk_ = true;
            } catch (const std::exception &) { remove_index_files(); }
        }
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

void Store::upsert_documents_lines(const std::vector<std::pair<std::string, std::vector<std::string_view>>> &docs, const search::StaticModel &model)
{
    if (docs.empty()) return;
    if (docs.size() == 1) { upsert_document_lines(docs[0].first, docs[0].second, model); return; }
    size_t total = 0;
    for (auto &d : docs) total += d.second.size();
    std::vector<std::string_view> all;
    all.reserve(total);
    for (auto &d : docs) {
        auto it = extents_.find(d.first);
        if (it != extents_.end()) dead_rows_ += it->second.n_rows;  // the whole old document is replaced (no stale tail)
        all.insert(all.end(), d.second.begin(), d.second.end());
    }
    const bool cache = token_cache_enabled();
    search::TokenCsr tokens;
    const auto t_embed = std::chrono::steady_clock::now();
    const std::function<void()> ahead = [this] { write_rows_ahead(); };
    uint64_t row = model.encode_into(all, 2048, 16384, corpus_, cache ? &tokens : nullptr, &ahead);
    search::PhaseTimer::add("within_persist:encode_into", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_embed).count());
    const auto t_tok = std::chrono::steady_clock::now();
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
    search::PhaseTimer::add("within_persist:token_log_append", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_tok).count());
}

// The rows embedded so far, beyond what the file holds, go to their places in line_embeddings.f32 NOW -- queued to the corpus'
// background writer (its own copy stream, page-cache writes and the device write-out off this thread) -- while the next batch is
// tokenised and pooled.  Not durable and not named by the
// header until flush_line_embeddings commits: a crash leaves the previous, consistent store.  Only in the append-only case (the file
// is the prefix of the corpus, or there is no store yet) and on one GPU; anything else keeps the rows for the flush.
void Store::write_rows_ahead() const
{
    static const bool off = [] { const char *e = getenv("SEMTOOLS_WRITE_AHEAD"); return e && e[0] == '0'; }();
    if (off) return;
    int n_ranks = 1;
    (void)smt_group_info(group_, &n_ranks, nullptr, nullptr, nullptr, nullptr);
    if (n_ranks != 1) return;
    const std::string emb = dir_ + "/line_embeddings.f32";
    const bool exists = path_exists(emb);
    if (exists && !rows_on_disk_valid_) return;                       // rows already in the file changed: the flush rewrites it
    if (!exists && (rows_on_disk_valid_ || rows_on_disk_ != 0)) return;
    const uint64_t rows = smt_sharded_corpus_rows(corpus_);
    const uint64_t written = std::max(rows_on_disk_, rows_written_ahead_);
    if (rows < rows_on_disk_ || rows <= written) return;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = smt_sharded_corpus_append_to_file_ex(corpus_, emb.c_str(), rows_on_disk_, written, SMT_APPEND_WRITE_AHEAD | (exists ? 0 : SMT_APPEND_CREATE));
    search::PhaseTimer::add("within_persist:rows_written_ahead", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (rc != SMT_OK) return;                                          // (the flush writes everything and reports what is wrong)
    rows_written_ahead_ = rows;
    if (!exists) rows_on_disk_valid_ = true;                           // an empty file now: the prefix (0 rows) of the corpus
}

void Store::upsert_document_lines(const std::string &path, const std::vector<std::string_view> &lines_for_embedding,
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
    // stat + read on up to eight threads (a repository is thousands of small files: open / read / close one by one was the whole
    // change-detection phase); states keep the order of file_paths, the first unreadable file IN THAT ORDER is the error raised
    // (store.rs:549-611 reads them in order with `?`)
    std::vector<std::optional<DocumentState>> found(file_paths.size());
    std::vector<std::exception_ptr> failed(file_paths.size());
    parallel_slices(file_paths.size(), 8, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const std::string &fp = file_paths[i];
            try {
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
                found[i] = std::move(ds);
            } catch (...) { failed[i] = std::current_exception(); }
        }
    });
    std::vector<DocumentState> states;
    for (size_t i = 0; i < file_paths.size(); ++i) {
        if (failed[i]) std::rethrow_exception(failed[i]);
        if (found[i]) states.push_back(std::move(*found[i]));
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
    const auto t_log = std::chrono::steady_clock::now();
    token_log_close();
    search::PhaseTimer::add("within_persist:token_log_close", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_log).count());
    // vectors first, then the extent table that references them (a crash in between leaves extra
    // rows that no extent points at -- harmless; the reverse order could reference missing rows)
    const std::string emb = dir_ + "/line_embeddings.f32";
    const uint64_t rows = smt_sharded_corpus_rows(corpus_);
    const auto t_rows = std::chrono::steady_clock::now();
    if (rows_on_disk_valid_ && rows >= rows_on_disk_ && path_exists(emb)) {
        // (rows written ahead while the later batches were embedded are only synced here; the header goes last)
        const uint64_t written = std::min(rows, std::max(rows_on_disk_, rows_written_ahead_));
        if (rows > rows_on_disk_) check(smt_sharded_corpus_append_to_file_ex(corpus_, emb.c_str(), rows_on_disk_, written, 0), "flush_line_embeddings");
    } else {
        check(smt_sharded_corpus_save(corpus_, emb.c_str()), "flush_line_embeddings");  // first flush or after a compaction
    }
    rows_written_ahead_ = 0;
    search::PhaseTimer::add("within_persist:rows_file", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_rows).count());
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
    root.set("generation", json::Value::uint(generation_));
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

}  // namespace semtools
