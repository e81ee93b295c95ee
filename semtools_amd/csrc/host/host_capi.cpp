// host_capi.cpp -- extern "C" surface of the host layer (include/semtools_host.h).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <unistd.h>

#include "../../../include/semtools_host.h"
#include "../common.h"
#include "host.h"
#include "host_internal.h"
#include "json.h"
#include "fmt.h"

using namespace semtools;

// The host layer runs on a group (host.h).  Entry points that take an smt_ctx wrap it in a one-rank group, for which
// every sharded call is its single-GPU counterpart.
struct GroupRef {
    smt_group *g = nullptr;
    bool owned = false;   // a one-rank group made around the caller's context (the context itself stays the caller's)
    GroupRef() = default;
    GroupRef(const GroupRef &) = delete;
    GroupRef &operator=(const GroupRef &) = delete;
    ~GroupRef() { if (owned && g) smt_group_destroy(g); }
    void wrap(smt_ctx *ctx)
    {
        if (smt_group_from_ctx(ctx, &g) != SMT_OK) throw semtools::Error(smt_last_error());
        owned = true;
    }
    void use(smt_group *group) { g = group; owned = false; }
};

struct smt_host_model {
    GroupRef group;                          // (declared first: destroyed after the model that lives on it)
    std::unique_ptr<search::StaticModel> m;
};

struct smt_host_session {
    smt_host_model *model = nullptr;
    std::unique_ptr<search::Embeddings> emb;
    std::vector<search::Document> docs;
    bool ignore_case = false;
};

namespace {

int fail(const std::exception &e)
{
    smt::set_error("%s", e.what());
    return SMT_E_INVALID;
}

char *dup_text(const std::string &s)
{
    char *p = (char *)malloc(s.size() + 1);
    if (!p) return nullptr;
    memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    return p;
}

search::SearchConfig make_config(uint64_t n_lines, uint64_t top_k, double max_distance, int ignore_case)
{
    search::SearchConfig c;
    c.n_lines = (size_t)n_lines;
    c.top_k = (size_t)top_k;
    if (!std::isnan(max_distance)) c.max_distance = max_distance;
    c.ignore_case = ignore_case != 0;
    return c;
}

// ---- minimal safetensors reader: u64 header length, JSON header, raw little-endian data
std::vector<float> read_safetensors_embeddings(const std::string &path, uint64_t &V)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error("cannot open " + path);
    uint64_t hlen = 0;
    f.read(reinterpret_cast<char *>(&hlen), 8);
    if (!f || hlen > (1ull << 30)) throw Error("bad safetensors header in " + path);
    std::string hdr(hlen, '\0');
    f.read(&hdr[0], (std::streamsize)hlen);
    const json::Value h = json::parse(hdr);
    const json::Value *t = h.get("embeddings");
    if (!t) throw Error("tensor 'embeddings' not found in " + path);
    const std::string dtype = t->get("dtype")->s;
    const auto &shape = t->get("shape")->arr;
    if (shape.size() != 2 || shape[1].as_u64() != SMT_DIM) throw Error("'embeddings' must be [V, 256]");
    V = shape[0].as_u64();
    const uint64_t b0 = t->get("data_offsets")->arr[0].as_u64();
    std::vector<float> out((size_t)V * SMT_DIM);
    f.seekg((std::streamoff)(8 + hlen + b0));
    if (dtype == "F32") {
        f.read(reinterpret_cast<char *>(out.data()), (std::streamsize)(out.size() * 4));
    } else if (dtype == "F16") {
        std::vector<uint16_t> raw(out.size());
        f.read(reinterpret_cast<char *>(raw.data()), (std::streamsize)(raw.size() * 2));
        for (size_t i = 0; i < raw.size(); ++i) {  // IEEE half -> float
            const uint32_t s = (raw[i] >> 15) & 1, e = (raw[i] >> 10) & 0x1F, m = raw[i] & 0x3FF;
            uint32_t bits;
            if (e == 0) {
                if (m == 0) bits = s << 31;
                else { int sh = 0; uint32_t mm = m; while (!(mm & 0x400)) { mm <<= 1; ++sh; }
                       bits = (s << 31) | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 0x3FF) << 13); }
            } else if (e == 31) bits = (s << 31) | 0x7F800000u | (m << 13);
            else bits = (s << 31) | ((e - 15 + 127) << 23) | (m << 13);
            memcpy(&out[i], &bits, 4);
        }
    } else if (dtype == "I8") {  // model2vec-rs converts each byte `as i8 as f32` (no scale)
        std::vector<int8_t> raw(out.size());
        f.read(reinterpret_cast<char *>(raw.data()), (std::streamsize)raw.size());
        for (size_t i = 0; i < raw.size(); ++i) out[i] = (float)raw[i];
    } else throw Error("unsupported embeddings dtype " + dtype + " (F32, F16, I8)");
    if (!f) throw Error("truncated safetensors file " + path);
    return out;
}

// dtype / shape / byte offset of the `embeddings` tensor (no data read)
bool safetensors_f32_span(const std::string &path, uint64_t &V, uint64_t &byte_offset)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error("cannot open " + path);
    uint64_t hlen = 0;
    f.read(reinterpret_cast<char *>(&hlen), 8);
    if (!f || hlen > (1ull << 30)) throw Error("bad safetensors header in " + path);
    std::string hdr(hlen, '\0');
    f.read(&hdr[0], (std::streamsize)hlen);
    const json::Value h = json::parse(hdr);
    const json::Value *t = h.get("embeddings");
    if (!t) throw Error("tensor 'embeddings' not found in " + path);
    const auto &shape = t->get("shape")->arr;
    if (shape.size() != 2 || shape[1].as_u64() != SMT_DIM) throw Error("'embeddings' must be [V, 256]");
    V = shape[0].as_u64();
    byte_offset = 8 + hlen + t->get("data_offsets")->arr[0].as_u64();
    return t->get("dtype")->s == "F32";
}

}  // namespace

extern "C" {

static int host_model_create(smt_ctx *ctx, smt_group *group, const float *table, uint64_t V, int normalize, int tok_kind,
                             const char *vocab_path, const char *unk_token, smt_tokenize_cb cb, void *user, uint32_t unk_id,
                             uint32_t median_len, smt_host_model **out)
{
    if ((!ctx && !group) || !table || !out) { smt::set_error("null argument"); return SMT_E_INVALID; }
    *out = nullptr;
    try {
        std::unique_ptr<Tokenizer> tok;
        if (tok_kind == SMT_TOK_HASH) tok = make_hash_tokenizer(V);
        else if (tok_kind == SMT_TOK_VOCAB) tok = make_vocab_tokenizer(vocab_path ? vocab_path : "", unk_token ? unk_token : "");
        else if (tok_kind == SMT_TOK_CALLBACK) {
            if (!cb) { smt::set_error("callback tokenizer without a callback"); return SMT_E_INVALID; }
            TokenizeFn fn = [cb, user](const std::string &text, std::vector<uint32_t> &ids) {
                uint64_t n = 0;
                ids.resize(std::max<size_t>(64, text.size() + 8));
                if (cb(user, text.data(), text.size(), ids.data(), ids.size(), &n) != 0) throw Error("tokenizer callback failed");
                if (n > ids.size()) {
                    ids.resize(n);
                    if (cb(user, text.data(), text.size(), ids.data(), ids.size(), &n) != 0) throw Error("tokenizer callback failed");
                }
                ids.resize(n);
            };
            tok = make_callback_tokenizer(std::move(fn), V, unk_id == UINT32_MAX ? std::nullopt : std::optional<uint32_t>(unk_id),
                                          median_len ? median_len : 5);
        } else { smt::set_error("unknown tokenizer kind %d", tok_kind); return SMT_E_INVALID; }
        std::unique_ptr<smt_host_model> h(new smt_host_model());
        if (group) h->group.use(group); else h->group.wrap(ctx);
        h->m = std::make_unique<search::StaticModel>(h->group.g, std::move(tok), table, V, normalize != 0);
        *out = h.release();
        return SMT_OK;
    } catch (const std::exception &e) { return fail(e); }
}

int smt_host_model_create(smt_ctx *ctx, const float *table, uint64_t V, int normalize, int tok_kind, const char *vocab_path,
                          const char *unk_token, smt_tokenize_cb cb, void *user, uint32_t unk_id, uint32_t median_len,
                          smt_host_model **out)
{
    return host_model_create(ctx, nullptr, table, V, normalize, tok_kind, vocab_path, unk_token, cb, user, unk_id, median_len, out);
}

int smt_host_model_create_group(smt_group *group, const float *table, uint64_t V, int normalize, int tok_kind, const char *vocab_path,
                                const char *unk_token, smt_tokenize_cb cb, void *user, uint32_t unk_id, uint32_t median_len,
                                smt_host_model **out)
{
    return host_model_create(nullptr, group, table, V, normalize, tok_kind, vocab_path, unk_token, cb, user, unk_id, median_len, out);
}

// "0,1,2" = those GPUs (RCCL between them); "all" = every visible GPU; "<d>:<n>" = n logical shards on GPU d (a test rig
// with one GPU: device copies stand in for RCCL); "<d>" = GPU d alone.
int smt_host_group_from_spec(const char *spec, smt_group **out)
{
    if (!spec || !out) { smt::set_error("null argument"); return SMT_E_INVALID; }
    *out = nullptr;
    const std::string s(spec);
    if (s == "all") {
        const int n = smt_device_count();
        if (n <= 0) { if (n == 0) smt::set_error("no HIP device visible: libsemtools_hip has no CPU fallback"); return SMT_E_HIP; }
        if (n == 1) return smt_group_create_logical(0, 1, out);
        std::vector<int> devs;
        for (int i = 0; i < n; ++i) devs.push_back(i);
        return smt_group_create(devs.data(), n, out);
    }
    const size_t colon = s.find(':');
    char *end = nullptr;
    if (colon != std::string::npos) {
        const long dev = strtol(s.c_str(), &end, 10);
        const long n = strtol(s.c_str() + colon + 1, nullptr, 10);
        if (end != s.c_str() + colon || dev < 0 || n < 1) { smt::set_error("device spec '%s': expected <device>:<shards>", spec); return SMT_E_INVALID; }
        return smt_group_create_logical((int)dev, (int)n, out);
    }
    std::vector<int> devs;
    for (const char *p = s.c_str(); *p;) {
        const long d = strtol(p, &end, 10);
        if (end == p || d < 0 || (*end && *end != ',')) { smt::set_error("device spec '%s': expected a comma-separated list of GPU ordinals", spec); return SMT_E_INVALID; }
        devs.push_back((int)d);
        p = *end ? end + 1 : end;
    }
    if (devs.empty()) { smt::set_error("device spec is empty"); return SMT_E_INVALID; }
    if (devs.size() == 1) return smt_group_create_logical(devs[0], 1, out);   // one GPU: no communicator needed
    return smt_group_create(devs.data(), (int)devs.size(), out);
}

char *smt_host_timing_json(void) { return dup_text(search::PhaseTimer::json()); }

struct smt_host_tokenizer {
    std::unique_ptr<Tokenizer> t;
};

int smt_host_tokenizer_load(const char *tokenizer_json_path, smt_host_tokenizer **out)
{
    if (!tokenizer_json_path || !out) { smt::set_error("null argument"); return SMT_E_INVALID; }
    *out = nullptr;
    try {
        auto *h = new smt_host_tokenizer();
        h->t = make_hf_tokenizer(tokenizer_json_path);
        *out = h;
        return SMT_OK;
    } catch (const std::exception &e) { return fail(e); }
}

void smt_host_tokenizer_free(smt_host_tokenizer *tok) { delete tok; }

int smt_host_tokenizer_encode(smt_host_tokenizer *tok, const char *text, uint32_t *ids, uint64_t cap, uint64_t *n_ids)
{
    if (!tok || !text || !n_ids || (cap && !ids)) { smt::set_error("null argument"); return SMT_E_INVALID; }
    try {
        std::vector<uint32_t> v;
        tok->t->encode(text, v);
        *n_ids = v.size();
        for (uint64_t i = 0; i < std::min<uint64_t>(cap, v.size()); ++i) ids[i] = v[i];
        return v.size() > cap ? SMT_E_TRUNCATED : SMT_OK;
    } catch (const std::exception &e) { return fail(e); }
}

int smt_host_tokenizer_info(smt_host_tokenizer *tok, uint64_t *vocab_size, int64_t *unk_id, uint64_t *median_token_bytes)
{
    if (!tok) { smt::set_error("null argument"); return SMT_E_INVALID; }
    if (vocab_size) *vocab_size = tok->t->vocab_size();
    if (unk_id) *unk_id = tok->t->unk_id() ? (int64_t)*tok->t->unk_id() : -1;
    if (median_token_bytes) *median_token_bytes = tok->t->median_token_length();
    return SMT_OK;
}

static int host_model_from_dir(smt_ctx *ctx, smt_group *group, const char *dir, smt_host_model **out)
{
    if ((!ctx && !group) || !dir || !out) { smt::set_error("null argument"); return SMT_E_INVALID; }
    *out = nullptr;
    try {
        const std::string d(dir);
        search::PhaseTimer::mark("process_start_and_hip_context");   // (everything before the model directory is touched)
        uint64_t V = 0, table_offset = 0;
        // F32 tables (what model2vec ships) stream file -> pinned -> HBM; F16 / I8 tables are widened on the host first
        const bool stream_f32 = safetensors_f32_span(d + "/model.safetensors", V, table_offset);
        std::vector<float> table;
        if (!stream_f32) table = read_safetensors_embeddings(d + "/model.safetensors", V);
        bool normalize = true;
        std::string unk = "[UNK]";
        try {
            const json::Value cfg = json::parse(read_to_string(d + "/config.json"));
            if (auto *x = cfg.get("normalize")) normalize = x->b;
            if (auto *x = cfg.get("unk_token")) unk = x->s;
        } catch (const std::exception &) {}
        // a real model2vec directory carries tokenizer.json (read natively: hf_tokenizer.cpp); the synthetic test
        // models carry a plain vocab.txt
        std::unique_ptr<Tokenizer> tok;
        {
            std::ifstream tj(d + "/tokenizer.json");
            if (tj.good()) tok = make_hf_tokenizer(d + "/tokenizer.json");
            else tok = make_vocab_tokenizer(d + "/vocab.txt", unk);
        }
        if (tok->vocab_size() > V) throw Error("vocab.txt has more tokens than the embedding table has rows");
        search::PhaseTimer::mark("tokenizer_load");
        std::unique_ptr<smt_host_model> h(new smt_host_model());
        if (group) h->group.use(group); else h->group.wrap(ctx);
        if (stream_f32) h->m = std::make_unique<search::StaticModel>(h->group.g, std::move(tok), d + "/model.safetensors", table_offset, V, normalize);
        else h->m = std::make_unique<search::StaticModel>(h->group.g, std::move(tok), table.data(), V, normalize);
        *out = h.release();
        return SMT_OK;
    } catch (const std::exception &e) { return fail(e); }
}

int smt_host_model_from_dir(smt_ctx *ctx, const char *dir, smt_host_model **out) { return host_model_from_dir(ctx, nullptr, dir, out); }
int smt_host_model_from_dir_group(smt_group *group, const char *dir, smt_host_model **out) { return host_model_from_dir(nullptr, group, dir, out); }

void smt_host_model_destroy(smt_host_model *model) { delete model; }

int smt_host_encode(smt_host_model *model, const char *const *texts, uint64_t n, uint32_t max_length, float *out)
{
    if (!model || (n && (!texts || !out))) { smt::set_error("null argument"); return SMT_E_INVALID; }
    try {
        std::vector<std::string> s(texts, texts + n);
        auto v = model->m->encode_with_args(s, max_length ? std::optional<size_t>(max_length) : std::nullopt, 16384);
        for (uint64_t i = 0; i < n; ++i) memcpy(out + i * SMT_DIM, v[i].data(), SMT_DIM * sizeof(float));
        return SMT_OK;
    } catch (const std::exception &e) { return fail(e); }
}

int smt_host_search_files(smt_host_model *model, const char *query, const char *const *files, uint64_t n_files, uint64_t n_lines,
                          uint64_t top_k, double max_distance, int ignore_case, int json, int is_tty, char **out_text)
{
    if (!model || !query || !out_text || (n_files && !files)) { smt::set_error("null argument"); return SMT_E_INVALID; }
    *out_text = nullptr;
    try {
        search::PhaseTimer::mark("between_calls");   // (a library caller's own time since the last phase is not split_lines)
        const auto cfg = make_config(n_lines, top_k, max_distance, ignore_case);
        const std::string q = ignore_case ? to_lowercase(query) : std::string(query);  // src/cmds/search.rs:130-134
        const auto res = search::search_files(std::vector<std::string>(files, files + n_files), q, *model->m, cfg);
        *out_text = dup_text(json ? cmds::search_results_json(res) : cmds::print_search_results(res, is_tty != 0));
        return SMT_OK;
    } catch (const std::exception &e) { return fail(e); }
}

int smt_host_search_content(smt_host_model *model, const char *query, const char *filename, const char *content, uint64_t n_lines,
                            uint64_t top_k, double max_distance, int ignore_case, int json, int is_tty, char **out_text)
{
    if (!model || !query || !content || !out_text) { smt::set_error("null argument"); return SMT_E_INVALID; }
    *out_text = nullptr;
    try {
        search::PhaseTimer::mark("between_calls");   // (a library caller's own time since the last phase is not split_lines)
        const auto cfg = make_config(n_lines, top_k, max_distance, ignore_case);
        const std::string q = ignore_case ? to_lowercase(query) : std::string(query);
        {
            search::Embeddings emb(model->m->group());
            std::vector<search::Document> docs;
            auto doc = search::create_document_from_content(filename ? filename : "<stdin>", std::string_view(content), *model->m, ignore_case != 0, emb);
            if (doc) docs.push_back(std::move(*doc));
            const auto res = search::search_documents(docs, emb, model->m->encode_single(q), cfg);
            *out_text = dup_text(json ? cmds::search_results_json(res) : cmds::print_search_results(res, is_tty != 0));
            search::PhaseTimer::mark("format_output");
        }
        search::PhaseTimer::mark("release_lines_and_rows");
        return SMT_OK;
    } catch (const std::exception &e) { return fail(e); }
}

int smt_host_search_workspace(smt_host_model *model, const char *query, const char *const *files, uint64_t n_files, uint64_t n_lines,
                              uint64_t top_k, double max_distance, int ignore_case, const char *workspace_name, int json, int is_tty,
                              char **out_text)
{
    if (!model || !query || !out_text || (n_files && !files)) { smt::set_error("null argument"); return SMT_E_INVALID; }
    *out_text = nullptr;
    try {
        search::PhaseTimer::mark("between_calls");   // (a library caller's own time since the last phase is not split_lines)
        const auto cfg = make_config(n_lines, top_k, max_distance, ignore_case);
        const std::string q = ignore_case ? to_lowercase(query) : std::string(query);
        std::optional<std::string> ws;
        if (workspace_name) ws = workspace_name;
        const auto ranked = search::search_with_workspace(std::vector<std::string>(files, files + n_files), q, *model->m, cfg, ws);
        *out_text = dup_text(json ? cmds::workspace_results_json(ranked, (size_t)n_lines)
                                  : cmds::print_workspace_search_results(ranked, (size_t)n_lines, is_tty != 0));
        return SMT_OK;
    } catch (const std::exception &e) { return fail(e); }
}

int smt_host_session_open(smt_host_model *model, const char *const *files, uint64_t n_files, int ignore_case, smt_host_session **out)
{
    if (!model || !out || (n_files && !files)) { smt::set_error("null argument"); return SMT_E_INVALID; }
    *out = nullptr;
    try {
        std::unique_ptr<smt_host_session> s(new smt_host_session());
        s->model = model;
        s->ignore_case = ignore_case != 0;
        s->emb = std::make_unique<search::Embeddings>(model->m->group());
        // (one embedding pipeline run over all files, like search_files: a repository is thousands of small files)
        s->docs = search::load_documents(std::vector<std::string>(files, files + n_files), *model->m, s->ignore_case, *s->emb);
        *out = s.release();
        return SMT_OK;
    } catch (const std::exception &e) { return fail(e); }
}

int smt_host_session_search(smt_host_session *s, const char *const *queries, uint64_t n_queries, uint64_t n_lines, uint64_t top_k,
                            double max_distance, int json, int is_tty, char **out_texts)
{
    if (!s || (n_queries && (!queries || !out_texts))) { smt::set_error("null argument"); return SMT_E_INVALID; }
    for (uint64_t i = 0; i < n_queries; ++i) out_texts[i] = nullptr;
    try {
        const auto cfg = make_config(n_lines, top_k, max_distance, s->ignore_case);
        std::vector<std::string> qs;
        for (uint64_t i = 0; i < n_queries; ++i) qs.push_back(s->ignore_case ? to_lowercase(queries[i]) : std::string(queries[i]));
        search::PhaseTimer::mark("between_session_calls");
        const auto qemb = s->model->m->encode_with_args(qs, 512, 1024);  // encode_single per query
        search::PhaseTimer::mark("session_encode_queries");
        const auto res = search::search_documents_batch(s->docs, *s->emb, qemb, cfg);
        search::PhaseTimer::mark("session_search_and_build_results");
        semtools::parallel_slices((size_t)n_queries, 128, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i)
                out_texts[i] = dup_text(json ? cmds::search_results_json(res[i]) : cmds::print_search_results(res[i], is_tty != 0));
        });
        search::PhaseTimer::mark("session_format");
        return SMT_OK;
    } catch (const std::exception &e) { return fail(e); }
}

uint64_t smt_host_session_lines(const smt_host_session *s) { return s ? s->emb->rows() : 0; }
void smt_host_session_close(smt_host_session *s) { delete s; }

static int host_workspace_use(smt_ctx *ctx, smt_group *group, const char *name, int json_out, char **out_text)
{
    if (!name || !out_text) { smt::set_error("null argument"); return SMT_E_INVALID; }
    *out_text = nullptr;
    try {
        GroupRef gr;
        if (group) gr.use(group); else if (ctx) gr.wrap(ctx);
        workspace::Workspace ws;
        ws.config.name = name;
        ws.config.root_dir = workspace::Workspace::root_path(name);
        ws.save();
        std::string out;
        if (json_out) {
            size_t total = 0;
            if (gr.g) { try { total = workspace::Store::open(ws.config.root_dir, gr.g)->get_stats().total_documents; } catch (const std::exception &) {} }
            json::Value o = json::Value::object();  // WorkspaceOutput (src/json_mode.rs:41-46)
            o.set("name", json::Value::str(ws.config.name));
            o.set("root_dir", json::Value::str(ws.config.root_dir));
            o.set("total_documents", json::Value::uint(total));
            out = json::to_string_pretty(o) + "\n";
        } else {
            const std::string n(name);  // src/cmds/workspace.rs:46-53
            out = "Workspace '" + n + "' configured.\nTo activate it, run:\n  export SEMTOOLS_WORKSPACE=" + n +
                  "\n\nOr add this to your shell profile (.bashrc, .zshrc, etc.)\n\nOr use the `--workspace` option on the commands that support it\n";
        }
        *out_text = dup_text(out);
        return SMT_OK;
    } catch (const std::exception &e) { return fail(e); }
}

int smt_host_workspace_use(smt_ctx *ctx, const char *name, int json_out, char **out_text) { return host_workspace_use(ctx, nullptr, name, json_out, out_text); }
int smt_host_workspace_use_group(smt_group *group, const char *name, int json_out, char **out_text) { return host_workspace_use(nullptr, group, name, json_out, out_text); }

static int host_workspace_status(smt_ctx *ctx, smt_group *group, const char *name_or_null, int json_out, char **out_text)
{
    if ((!ctx && !group) || !out_text) { smt::set_error("null argument"); return SMT_E_INVALID; }
    *out_text = nullptr;
    try {
        GroupRef gr;
        if (group) gr.use(group); else gr.wrap(ctx);
        std::optional<std::string> nm;
        if (name_or_null) nm = name_or_null;
        try { workspace::Workspace::active(nm); } catch (const Error &) { throw Error("No active workspace"); }
        const auto ws = workspace::Workspace::open(nm);
        const auto stats = workspace::Store::open(ws.config.root_dir, gr.g)->get_stats();
        std::string out;
        if (json_out) {
            json::Value o = json::Value::object();
            o.set("name", json::Value::str(ws.config.name));
            o.set("root_dir", json::Value::str(ws.config.root_dir));
            o.set("total_documents", json::Value::uint(stats.total_documents));
            out = json::to_string_pretty(o) + "\n";
        } else {  // src/cmds/workspace.rs:87-96
            out = "Active workspace: " + ws.config.name + "\nRoot: " + ws.config.root_dir + "\nDocuments: " +
                  std::to_string(stats.total_documents) + "\n";
            out += stats.has_index ? "Index: Yes (" + stats.index_type.value_or("Unknown") + ")\n" : "Index: No\n";
        }
        *out_text = dup_text(out);
        return SMT_OK;
    } catch (const std::exception &e) { return fail(e); }
}

int smt_host_workspace_status(smt_ctx *ctx, const char *name_or_null, int json_out, char **out_text) { return host_workspace_status(ctx, nullptr, name_or_null, json_out, out_text); }
int smt_host_workspace_status_group(smt_group *group, const char *name_or_null, int json_out, char **out_text) { return host_workspace_status(nullptr, group, name_or_null, json_out, out_text); }

int smt_host_workspace_reembed(smt_host_model *model, const char *name_or_null, int json_out, char **out_text)
{
    if (!model || !out_text) { smt::set_error("null argument"); return SMT_E_INVALID; }
    *out_text = nullptr;
    try {
        std::optional<std::string> nm;
        if (name_or_null) nm = name_or_null;
        try { workspace::Workspace::active(nm); } catch (const Error &) { throw Error("No active workspace"); }
        const auto ws = workspace::Workspace::open(nm);
        auto store = workspace::Store::open(ws.config.root_dir, model->m->group());
        const auto rep = store->reembed_from_token_cache(*model->m);
        std::string out;
        if (json_out) {
            json::Value o = json::Value::object();
            o.set("documents_reembedded", json::Value::uint(rep.documents));
            o.set("lines_reembedded", json::Value::uint(rep.lines));
            o.set("tokens_pooled", json::Value::uint(rep.tokens));
            json::Value miss = json::Value::array();
            for (auto &p : rep.missing) miss.arr.push_back(json::Value::str(p));
            o.set("documents_without_cached_tokens", std::move(miss));
            out = json::to_string_pretty(o) + "\n";
        } else if (!rep.missing.empty()) {
            out = "No cached tokens for " + std::to_string(rep.missing.size()) + " documents (nothing was changed):\n";
            for (auto &p : rep.missing) out += "  - " + p + "\n";
            out += "Search them once with this model to re-embed them from their files.\n";
        } else {
            out = "Re-embedded " + std::to_string(rep.lines) + " lines of " + std::to_string(rep.documents) +
                  " documents from cached tokens (" + std::to_string(rep.tokens) + " tokens).\n";
        }
        *out_text = dup_text(out);
        return SMT_OK;
    } catch (const std::exception &e) { return fail(e); }
}

static int host_workspace_prune(smt_ctx *ctx, smt_group *group, const char *name_or_null, int json_out, char **out_text)
{
    if ((!ctx && !group) || !out_text) { smt::set_error("null argument"); return SMT_E_INVALID; }
    *out_text = nullptr;
    try {
        GroupRef gr;
        if (group) gr.use(group); else gr.wrap(ctx);
        std::optional<std::string> nm;
        if (name_or_null) nm = name_or_null;
        try { workspace::Workspace::active(nm); } catch (const Error &) { throw Error("No active workspace"); }
        const auto ws = workspace::Workspace::open(nm);
        auto store = workspace::Store::open(ws.config.root_dir, gr.g);
        const auto all_paths = store->get_all_document_paths();
        std::vector<std::string> missing;
        for (auto &p : all_paths) if (access(p.c_str(), F_OK) != 0) missing.push_back(p);  // Path::exists()
        const size_t removed = missing.size(), remaining = all_paths.size() - removed;
        if (!missing.empty()) store->delete_documents(missing);
        std::string out;
        if (json_out) {
            json::Value o = json::Value::object();  // PruneOutput (src/json_mode.rs:48-52)
            o.set("files_removed", json::Value::uint(removed));
            o.set("files_remaining", json::Value::uint(remaining));
            out = json::to_string_pretty(o) + "\n";
        } else if (missing.empty()) {
            out = "No stale documents found. Workspace is clean.\n";
        } else {  // src/cmds/workspace.rs:148-157
            out = "Found " + std::to_string(removed) + " stale documents:\n";
            for (auto &p : missing) out += "  - " + p + "\n";
            out += "Removed " + std::to_string(removed) + " stale documents from workspace.\n";
        }
        *out_text = dup_text(out);
        return SMT_OK;
    } catch (const std::exception &e) { return fail(e); }
}

int smt_host_workspace_prune(smt_ctx *ctx, const char *name_or_null, int json_out, char **out_text) { return host_workspace_prune(ctx, nullptr, name_or_null, json_out, out_text); }
int smt_host_workspace_prune_group(smt_group *group, const char *name_or_null, int json_out, char **out_text) { return host_workspace_prune(nullptr, group, name_or_null, json_out, out_text); }

void smt_host_free(char *text) { free(text); }

char *smt_host_format_float(double value, int mode)
{
    if (mode == 0) return dup_text(fmt::rust_display(value));
    if (mode == 1) return dup_text(fmt::rust_display((float)value));
    return dup_text(fmt::json_f64(value));
}

char *smt_host_split_lines(const char *content)
{
    const auto lines = lines_of(content ? content : "");
    std::string out = std::to_string(lines.size());  // "<count>\x1f<line>\x1f<line>..."
    for (auto &l : lines) { out.push_back('\x1f'); out += l; }
    return dup_text(out);
}

char *smt_host_to_lowercase(const char *text) { return dup_text(to_lowercase(text ? text : "")); }

}  // extern "C"
