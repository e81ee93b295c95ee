// host.h -- C++ host layer above the C ABI: the reference's search module
// (src/search/mod.rs) and workspace store (src/workspace/{mod,store}.rs) with
// the SAME names, argument meaning and error behaviour, re-implemented as thin
// callers of libsemtools_hip (embeddings never live on the host: a Document
// records which corpus rows are its lines).
//
// Everything here runs on an smt_group -- the GPUs the one calling process owns (src/bin/semtools.rs:134-135 is one
// synchronous task): the embedding table is replicated per GPU, every matrix of line embeddings is an
// smt_sharded_corpus whose rows are dealt over the GPUs, searches end in the all-gather + merge of group.cpp.  The
// default group has ONE rank (smt_group_from_ctx), for which every smt_sharded_* call IS its single-GPU counterpart.
//
// The reference is Rust; no Rust toolchain exists here, so this C++ layer is
// what the CLI replica and the tests drive.  A Rust maintainer would keep the
// reference's own host code and call the C ABI as INTEGRATION.md shows.
#pragma once
#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../../include/semtools_hip.h"

namespace semtools {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Rust str::lines(): split on '\n', strip one trailing '\r', no trailing empty piece.
// line_views: the lines as views into `content` (no allocation per line); lines_of: the same lines as owned strings.
std::vector<std::string_view> line_views(std::string_view content);
std::vector<std::string> lines_of(std::string_view content);
// Rust str::to_lowercase() (simple per-code-point mapping; see DESIGN.md for the caveat).
std::string to_lowercase(const std::string &s);
std::string read_to_string(const std::string &path);  // throws Error like `?` on io::Error

// ------------------------------------------------------------------ tokenizer
// Stand-in for tokenizers::Tokenizer (the HF crate the reference uses through
// model2vec-rs).  encode() returns ids with add_special_tokens=false; the
// model drops unk ids and truncates, exactly where model2vec-rs does.
class Tokenizer {
public:
    virtual ~Tokenizer() = default;
    virtual void encode(const std::string &text, std::vector<uint32_t> &ids) const = 0;
    virtual std::optional<uint32_t> unk_id() const { return std::nullopt; }
    virtual size_t median_token_length() const { return 5; }
    virtual uint64_t vocab_size() const = 0;
    // lines worth starting one more thread for, in a batch (a thread costs ~30 us to start: ~0.15 us per hashed line, ~1-4 us per
    // line through a tokenizer.json pipeline)
    virtual size_t lines_per_thread() const { return 8192; }
};
// whitespace words looked up in a vocab file (one token per line, id = line index)
std::unique_ptr<Tokenizer> make_vocab_tokenizer(const std::string &vocab_path, const std::string &unk_token);
// whitespace words hashed with FNV-1a into [0, vocab_size) -- for synthetic tests
std::unique_ptr<Tokenizer> make_hash_tokenizer(uint64_t vocab_size);
// a Hugging Face tokenizer.json, natively (hf_tokenizer.cpp): BertNormalizer / BertPreTokenizer / WordPiece, Metaspace / Unigram, ...
std::unique_ptr<Tokenizer> make_hf_tokenizer(const std::string &tokenizer_json_path);
// caller-provided function (e.g. a binding of the real HF tokenizer)
using TokenizeFn = std::function<void(const std::string &, std::vector<uint32_t> &)>;
std::unique_ptr<Tokenizer> make_callback_tokenizer(TokenizeFn fn, uint64_t vocab_size, std::optional<uint32_t> unk,
                                                   size_t median_len);

namespace search {

constexpr const char *MODEL_NAME = "minishlab/potion-multilingual-128M";  // src/search/mod.rs:16

// model2vec_rs::model::StaticModel
// Wall-clock phases of one CLI invocation (SEMTOOLS_TIMING=1 prints them to stderr as one JSON line): where the
// time of `semtools search` goes when the scan itself takes 0.15 ms.
struct PhaseTimer {
    static void mark(const char *phase);   // closes the running phase under `phase`
    static void add(const char *phase, double ms);   // a span measured by the caller (may overlap the running phase: "within_*")
    static std::string json();             // {"phase": ms, ...} in order of first appearance
};

// The token ids StaticModel pooled for a run of lines (unk ids dropped, truncated to max_length): lens[i] ids per line.
// The workspace keeps them (line_tokens.log) so that a new embedding table for the same tokenizer can re-embed the
// whole store on the GPU without reading or tokenising a single source file.
struct TokenCsr {
    std::vector<uint32_t> ids;
    std::vector<uint32_t> lens;
};

class StaticModel {
public:
    // table: [V x 256] f32 host array (the `embeddings` tensor), uploaded once (to every GPU of the group).
    StaticModel(smt_group *group, std::unique_ptr<Tokenizer> tok, const float *table, uint64_t V, bool normalize);
    // the f32 table sits at `byte_offset` of `path` (model.safetensors).  LAZY: nothing is uploaded until an embed call
    // shows what it needs.  A one-shot CLI run (c1: 1000 lines; a warm workspace search: the query alone) touches a few
    // thousand of the 500 k rows: those rows are read from the file (pread, a few MB), uploaded as a compact table
    // with remapped ids, pooled and dropped -- same values, same order, bit-identical embeddings -- instead of
    // streaming 512 MB through pinned buffers first (0.2-0.3 s, most of the CLI's wall time).  A call with more than
    // 32768 lines, or whose ids cover more than 1/16 of the table, uploads the whole table once (and for good).
    // SEMTOOLS_EAGER_MODEL=1 restores the eager upload.
    StaticModel(smt_group *group, std::unique_ptr<Tokenizer> tok, const std::string &path, uint64_t byte_offset, uint64_t V,
                bool normalize);
    ~StaticModel();
    StaticModel(const StaticModel &) = delete;

    // encode_with_args(sentences, Some(max_length), batch_size): returns host vectors
    std::vector<std::vector<float>> encode_with_args(const std::vector<std::string> &sentences,
                                                     std::optional<size_t> max_length, size_t batch_size) const;
    // same, but the rows are appended to `corpus` (resident); returns the first new row
    // (sink, if given, receives the pooled token ids of every sentence in order)
    // (after_batch, if given, runs on the calling thread each time a batch's rows are resident in `corpus` -- while the next
    // batch is being tokenised: the workspace store writes the new rows ahead to its file there)
    uint64_t encode_into(const std::vector<std::string> &sentences, std::optional<size_t> max_length,
                         size_t batch_size, smt_sharded_corpus *corpus, TokenCsr *sink = nullptr,
                         const std::function<void()> *after_batch = nullptr) const;
    // (the sentences as views: the lines of a file are embedded where they lie in its content)
    uint64_t encode_into(const std::vector<std::string_view> &sentences, std::optional<size_t> max_length,
                         size_t batch_size, smt_sharded_corpus *corpus, TokenCsr *sink = nullptr,
                         const std::function<void()> *after_batch = nullptr) const;
    // pool step only: n_lines lines given as token CSR (already filtered / truncated) appended to `corpus`
    void embed_tokens_into(const uint32_t *ids, const uint64_t *offsets, uint64_t n_lines, smt_sharded_corpus *corpus) const;
    // identifies the tokenizer (vocab size, unk id, ids of a fixed probe text): cached tokens are only valid for it
    uint64_t tokenizer_fingerprint() const;
    // encode_single(text) = encode(&[text]) -> max_length 512, batch 1024
    std::vector<float> encode_single(const std::string &sentence) const;

    smt_group *group() const { return group_; }
    const Tokenizer &tokenizer() const { return *tok_; }

private:
    void tokenize_batch(const std::string_view *sentences, size_t begin, size_t end,
                        std::optional<size_t> max_length, std::vector<uint32_t> &ids,
                        std::vector<uint64_t> &offsets) const;
    // one batch of token CSR -> rows (host buffer and / or appended to a corpus) through the full or a compact table
    void embed_csr(const std::vector<uint32_t> &ids, const std::vector<uint64_t> &offsets, uint64_t n_lines, float *out_host,
                   smt_sharded_corpus *corpus) const;
    smt_sharded_model *full_model() const;   // uploads the whole table on first use (lazy mode)
    smt_group *group_;
    std::unique_ptr<Tokenizer> tok_;
    mutable smt_sharded_model *model_ = nullptr;
    // lazy mode (file-backed f32 table not uploaded yet)
    std::string lazy_path_;
    uint64_t lazy_offset_ = 0, lazy_V_ = 0;
    bool lazy_normalize_ = true;
    int lazy_fd_ = -1;
    mutable std::vector<uint32_t> lazy_slot_;   // id -> compact slot + 1 (0 = unseen), reset after every call
    mutable uint32_t lazy_calls_ = 0;           // compact-table calls so far: a long series of small calls gets the full table too
};

// src/search/mod.rs:18-22.  `embeddings: Vec<Vec<f32>>` became a row range of the resident corpus.
// `lines: Vec<String>` became views into the document's content: a million-line file is a million heap strings to build and
// to free otherwise -- two thirds of the wall time of embedding it (profiles/r04_ingest_phases.json).
struct Document {
    std::string filename;
    std::vector<std::string_view> lines;       // views into *text, or into a buffer the caller keeps alive as long as the document
    std::shared_ptr<const std::string> text;   // the content, when the document owns it
    uint64_t first_row = 0;  // rows [first_row, first_row + lines.size()) of the owning corpus
};

struct SearchConfig {  // src/search/mod.rs:32-38
    size_t n_lines = 0;
    size_t top_k = 0;
    std::optional<double> max_distance;
    bool ignore_case = false;
};

struct SearchResult {  // src/search/mod.rs:40-47
    std::string filename;
    std::vector<std::string> lines;
    size_t start = 0;
    size_t end = 0;
    size_t match_line = 0;
    double distance = 0.0;
};

// The resident embedding matrix that a set of Documents points into.
class Embeddings {
public:
    explicit Embeddings(smt_group *group);
    ~Embeddings();
    Embeddings(const Embeddings &) = delete;
    smt_sharded_corpus *corpus() const { return corpus_; }
    uint64_t rows() const;

private:
    smt_sharded_corpus *corpus_ = nullptr;
};

// src/search/mod.rs:49-75: None for empty content; original lines kept; lower-cased copy embedded.
// The document's lines are views into `content`: the buffer stays with the caller and must outlive the document (documents read
// from files own their content: load_documents).
std::optional<Document> create_document_from_content(const std::string &filename, std::string_view content,
                                                     const StaticModel &model, bool ignore_case, Embeddings &emb);

// The documents of `files` (mod.rs:128-134: empty files give none), their lines embedded into `emb` in ONE pipeline run across the
// file borders; throws on the first unreadable file.
std::vector<Document> load_documents(const std::vector<std::string> &files, const StaticModel &model, bool ignore_case, Embeddings &emb);

// src/search/mod.rs:77-120.  `documents` must be in the order their lines were embedded into `emb`.
std::vector<SearchResult> search_documents(const std::vector<Document> &documents, const Embeddings &emb,
                                           const std::vector<float> &query_embedding, const SearchConfig &config);

// Batched form (no reference counterpart: the reference answers one query per process).  Same semantics per
// query as search_documents; all queries share ONE pass over the corpus (K3 when there are >= 8 of them).
std::vector<std::vector<SearchResult>> search_documents_batch(const std::vector<Document> &documents, const Embeddings &emb,
                                                              const std::vector<std::vector<float>> &query_embeddings,
                                                              const SearchConfig &config);

// src/search/mod.rs:122-143 (first unreadable file aborts: throws Error)
std::vector<SearchResult> search_files(const std::vector<std::string> &files, const std::string &query,
                                       const StaticModel &model, const SearchConfig &config);

}  // namespace search

namespace workspace {

constexpr uint32_t CURRENT_EMBEDDING_VERSION = 2;  // src/workspace/store.rs:29-34
constexpr size_t LINE_EMBEDDING_SIZE = 256;         // src/workspace/store.rs:37

struct WorkspaceConfig {  // src/workspace/mod.rs:8-26
    std::string name = "default";
    std::string root_dir;
    size_t in_batch_size = 5000;
    size_t oversample_factor = 3;
    // not in the reference (ignored by it: serde skips unknown keys): opt into the approximate index for whole-workspace
    // searches over at least this many lines (0 = never; SEMTOOLS_INDEX_MIN_ROWS overrides)
    uint64_t approximate_index_min_rows = 0;
};

struct Workspace {  // src/workspace/mod.rs:28-101
    WorkspaceConfig config;
    static Workspace open(const std::optional<std::string> &workspace_name);
    void save() const;
    static std::string active(const std::optional<std::string> &workspace_name);  // throws "No active workspace..."
    static std::string active_path(const std::optional<std::string> &workspace_name);
    static std::string root_path(const std::string &name);
    static std::string config_path_for(const std::string &name);
};

struct DocMeta {  // src/workspace/store.rs:52-58
    std::string path;
    uint64_t size_bytes = 0;
    int64_t mtime = 0;
    uint32_t _version = CURRENT_EMBEDDING_VERSION;
    uint64_t id() const;  // fnv1a(path)  (:75-80)
};

struct DocumentInfo {  // src/search/mod.rs:24-30
    std::string filename;
    std::string content;
    DocMeta meta;
};

struct DocumentState {  // src/workspace/store.rs:60-65
    enum Kind { Unchanged, Changed, New } kind;
    std::string filename;  // Unchanged
    DocumentInfo info;     // Changed / New
};

struct LineEmbedding {  // src/workspace/store.rs:67-73 (embedding is a row of the resident corpus)
    std::string path;
    int32_t line_number = 0;
    std::vector<float> embedding;  // host copy, only used by upsert_line_embeddings(host form)
    uint64_t id() const;           // fnv1a(path || line LE)  (:82-89)
};

struct RankedLine {  // src/workspace/store.rs:91-96
    std::string path;
    int32_t line_number = 0;
    float distance = 0.f;
};

struct WorkspaceStats {  // src/workspace/store.rs:98-103
    size_t total_documents = 0;
    bool has_index = true;
    std::optional<std::string> index_type;
};

// Storage wrapper (src/workspace/store.rs:105-647).  The two Qdrant shards became:
//   <dir>/documents.json      doc metadata (path, size_bytes, mtime, _version)
//   <dir>/line_rows.bin       per document: path + (first_row, n_rows)  [lines of a doc are contiguous rows]
//   <dir>/line_embeddings.f32 the resident corpus matrix (smt_corpus_save format, rows in global order whatever the
//                             number of GPUs that wrote it)
// line_rows.json also records how the rows were dealt over the GPUs ("shards"); a store opened by a group of the same
// size restores that layout (per-shard index files stay valid), any other group re-cuts the matrix evenly.
class Store {
public:
    static std::unique_ptr<Store> open(const std::string &workspace_dir, smt_group *group);
    ~Store();

    std::unordered_map<std::string, DocMeta> get_existing_docs(const std::vector<std::string> &paths) const;
    void delete_document_metadata(const std::vector<std::string> &paths);
    void delete_line_embeddings(const std::vector<std::string> &paths);
    void delete_documents(const std::vector<std::string> &paths);
    void upsert_document_metadata(const std::vector<DocMeta> &metas);
    // host-vector form (reference signature) ...
    void upsert_line_embeddings(const std::vector<LineEmbedding> &line_embeddings);
    // ... and the resident form used by search_with_workspace: embed straight into the store's corpus
    void upsert_document_lines(const std::string &path, const std::vector<std::string_view> &lines_for_embedding,
                               const search::StaticModel &model);
    // the same for a batch of documents through ONE embedding pipeline run (tokenise || H2D || K1 across document
    // borders; one round of tokenizer threads per 65536 lines instead of one per file)
    // (the lines are views: into the documents' contents, or into the lowered copies the caller holds)
    void upsert_documents_lines(const std::vector<std::pair<std::string, std::vector<std::string_view>>> &docs, const search::StaticModel &model);
    WorkspaceStats get_stats() const;
    std::vector<std::string> get_all_document_paths() const;
    std::vector<RankedLine> search_line_embeddings(const std::vector<float> &query_vec,
                                                   const std::vector<std::string> &subset_paths, size_t top_k,
                                                   std::optional<float> max_distance) const;
    std::vector<DocumentState> analyze_document_states(const std::vector<std::string> &file_paths) const;
    size_t count_documents() const;
    size_t count_line_embeddings() const;
    void flush_documents() const;
    void flush_line_embeddings() const;
    // Rows of replaced / deleted documents stay behind until they exceed half the matrix; then live extents are
    // rewritten back to back (row order preserved).  Called after every batch of upserts and deletes.
    void compact_if_sparse();
    // Token cache.  upsert_document_lines appends the pooled token ids of every document it embeds to
    // <dir>/line_tokens.log (append-only records, the latest record of a path wins, deletions write tombstones;
    // SEMTOOLS_TOKEN_CACHE=0 turns it off).  reembed_from_token_cache re-creates every stored vector from those ids
    // with `model` -- a new embedding table behind the SAME tokenizer (fingerprint checked) -- on the GPU, in row
    // order, without touching the source files; documents without a usable record are reported and nothing changes.
    struct ReembedReport {
        uint64_t documents = 0, lines = 0, tokens = 0;
        std::vector<std::string> missing;   // documents whose tokens are not cached (re-embed them from their files)
    };
    ReembedReport reembed_from_token_cache(const search::StaticModel &model);
    // Approximate index policy -- OPT-IN (min_rows = UINT64_MAX, the default, never uses it: the reference's store always
    // searches with `exact: true`, src/workspace/store.rs:619,632).  When enabled (SEMTOOLS_INDEX_MIN_ROWS, or
    // "approximate_index_min_rows" in the workspace's config.json), whole-workspace searches (the path subset covers
    // every stored document) over at least `min_rows` rows go through an IVF index with per-list PCA codes (local_pca = 1) that lives
    // beside the vectors (`line_index.ivf`), is extended incrementally when rows are appended and rebuilt when rows
    // move (compaction) or the corpus has doubled.  oversample_factor (WorkspaceConfig, src/workspace/mod.rs:13,22 --
    // vestigial in the reference, whose store scans exactly) sets the re-score depth: 2 * top_k * oversample_factor
    // ADC candidates per probed list (at least 64) are re-scored against the full-precision rows.  Every returned
    // distance is exact; only membership is approximate.  Searches over a path subset, top_k > 24, or smaller
    // stores use the exact scan, and so does a search to which the index returns fewer than top_k live rows.
    void set_index_policy(size_t oversample_factor, uint64_t min_rows, uint32_t nprobe);
    bool has_index() const { return index_ != nullptr || index_on_disk_; }

private:
    Store() = default;
    struct Extent { uint64_t first_row = 0; uint64_t n_rows = 0; };
    std::string dir_;
    smt_group *group_ = nullptr;
    smt_sharded_corpus *corpus_ = nullptr;
    void drop_index();                           // index_ (points into corpus_) and its files: rows are about to move
    void remove_index_files() const;             // every line_index.* of the directory, whatever rank count wrote it
    uint64_t generation_ = 0;                    // bumped by drop_index; line_rows.json and line_index.gen carry it
    std::string index_file(int rank) const;
    std::map<std::string, DocMeta> docs_;        // documents shard
    std::map<std::string, Extent> extents_;      // path -> rows holding its lines (line i = first_row + i)
    uint64_t dead_rows_ = 0;                     // rows of deleted/replaced documents awaiting compaction
    mutable uint64_t rows_on_disk_ = 0;          // prefix of the corpus already in line_embeddings.f32
    mutable bool rows_on_disk_valid_ = false;
    // rows [rows_on_disk_, rows_written_ahead_) sit in the file already (written while later batches were embedded, not durable,
    // not named by its header): the next flush only writes what is missing, syncs and commits (store.rs:402-434 flushes as it goes)
    mutable uint64_t rows_written_ahead_ = 0;
    void write_rows_ahead() const;
    // approximate index (built / extended lazily by the first search that qualifies)
    bool ensure_index() const;
    mutable smt_sharded_ivfpq *index_ = nullptr;
    mutable bool index_on_disk_ = false;
    mutable uint64_t index_built_rows_ = 0;      // corpus rows when the quantisers were trained
    size_t oversample_factor_ = 3;
    void token_log_append(const std::string &path, const search::TokenCsr *tokens, uint64_t fingerprint) const;  // null = tombstone
    void token_log_close() const;
    mutable FILE *token_log_file_ = nullptr;       // open while a series of appends is under way
    mutable uint64_t token_log_fingerprint_ = 0;   // fingerprint in the log's header (0 = not read yet / no log)
    uint64_t index_min_rows_ = UINT64_MAX;   // opt-in (see set_index_policy)
    uint32_t index_nprobe_ = 16;
};

}  // namespace workspace

namespace search {
// src/search/mod.rs:146-216
std::vector<workspace::RankedLine> search_with_workspace(const std::vector<std::string> &files, const std::string &query,
                                                         const StaticModel &model, const SearchConfig &config,
                                                         const std::optional<std::string> &workspace_name);
}  // namespace search

// ------------------------------------------------------------------ output (src/cmds/search.rs, src/json_mode.rs)
namespace cmds {
std::string print_search_results(const std::vector<search::SearchResult> &results, bool is_tty);       // :35-63
std::string print_workspace_search_results(const std::vector<workspace::RankedLine> &ranked, size_t n_lines,
                                           bool is_tty);                                               // :66-110
std::string search_results_json(const std::vector<search::SearchResult> &results);                     // :23-32
std::string workspace_results_json(const std::vector<workspace::RankedLine> &ranked, size_t n_lines);  // :208-241
}  // namespace cmds

}  // namespace semtools
