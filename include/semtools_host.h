/*
 * semtools_host.h -- C ABI of the HOST layer of libsemtools_hip.so: the reference's
 * search module and workspace store re-implemented over the kernel-level ABI
 * (semtools_hip.h).  These entry points exist so that non-C++ callers (tests, the
 * CLI replica) can drive the same functions the reference exposes:
 *
 *   smt_host_model_*          model2vec_rs::StaticModel (from_pretrained / encode*)
 *   smt_host_search_files     search_files + print_search_results      (src/search/mod.rs:122-143,
 *                                                                        src/cmds/search.rs:35-63,245-257)
 *   smt_host_search_content   the stdin branch of search_cmd            (src/cmds/search.rs:145-176)
 *   smt_host_search_workspace search_with_workspace + workspace printing (src/search/mod.rs:146-216,
 *                                                                        src/cmds/search.rs:66-110,197-244)
 *   smt_host_workspace_*      workspace use / status / prune            (src/cmds/workspace.rs)
 *
 * Text comes back as a malloc'd, NUL-terminated UTF-8 string (*out_text) that the caller
 * releases with smt_host_free; it is byte-for-byte what the reference prints to stdout.
 * Progress lines go to stderr exactly where the reference prints them.
 */
#ifndef SEMTOOLS_HOST_H
#define SEMTOOLS_HOST_H

#include "semtools_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smt_host_model smt_host_model;

#define SMT_TOK_HASH 0     /* whitespace words hashed (FNV-1a) into [0, V): synthetic tests      */
#define SMT_TOK_VOCAB 1    /* whitespace words looked up in vocab_path (one token per line)      */
#define SMT_TOK_CALLBACK 2 /* caller-provided tokenizer (e.g. the real HF tokenizers binding)    */

/* Writes up to `cap` ids of `text` to `ids`, the total count to *n; returns 0 on success. */
typedef int (*smt_tokenize_cb)(void *user, const char *text, uint64_t len, uint32_t *ids, uint64_t cap,
                               uint64_t *n);

/* StaticModel::from_pretrained: `table` = the f32 `embeddings` tensor [V x 256] (host);
 * unk_id = UINT32_MAX when the tokenizer has no unk token; median_len = median vocabulary
 * token length in characters (used only for callback tokenizers). */
int smt_host_model_create(smt_ctx *ctx, const float *table, uint64_t V, int normalize, int tok_kind,
                          const char *vocab_path, const char *unk_token, smt_tokenize_cb cb, void *user,
                          uint32_t unk_id, uint32_t median_len, smt_host_model **out);
/* directory with model.safetensors ("embeddings", F32 or F16), vocab.txt, optional config.json */
int smt_host_model_from_dir(smt_ctx *ctx, const char *dir, smt_host_model **out);
void smt_host_model_destroy(smt_host_model *model);

/* encode_with_args(texts, Some(max_length) / None when 0, batch 16384) -> out [n x 256] */
int smt_host_encode(smt_host_model *model, const char *const *texts, uint64_t n, uint32_t max_length,
                    float *out);

/* max_distance: NaN = None.  json != 0: SearchOutput JSON (to_string_pretty) instead of text. */
int smt_host_search_files(smt_host_model *model, const char *query, const char *const *files, uint64_t n_files,
                          uint64_t n_lines, uint64_t top_k, double max_distance, int ignore_case, int json,
                          int is_tty, char **out_text);
int smt_host_search_content(smt_host_model *model, const char *query, const char *filename, const char *content,
                            uint64_t n_lines, uint64_t top_k, double max_distance, int ignore_case, int json,
                            int is_tty, char **out_text);
int smt_host_search_workspace(smt_host_model *model, const char *query, const char *const *files,
                              uint64_t n_files, uint64_t n_lines, uint64_t top_k, double max_distance,
                              int ignore_case, const char *workspace_name, int json, int is_tty,
                              char **out_text);

/* Resident session (no reference counterpart -- the reference reloads the model and
 * re-embeds the files for every query): the files are embedded ONCE, then any number of query batches are
 * answered against the resident corpus.  out_texts[i] (malloc'd, release each with smt_host_free) is
 * byte-for-byte what `semtools search <queries[i]> <files...>` prints. */
typedef struct smt_host_session smt_host_session;
int smt_host_session_open(smt_host_model *model, const char *const *files, uint64_t n_files, int ignore_case,
                          smt_host_session **out);
int smt_host_session_search(smt_host_session *session, const char *const *queries, uint64_t n_queries,
                            uint64_t n_lines, uint64_t top_k, double max_distance, int json, int is_tty,
                            char **out_texts);
uint64_t smt_host_session_lines(const smt_host_session *session);
void smt_host_session_close(smt_host_session *session);

/* workspace use / status / prune: same stdout text / JSON as src/cmds/workspace.rs */
int smt_host_workspace_use(smt_ctx *ctx, const char *name, int json, char **out_text);
int smt_host_workspace_status(smt_ctx *ctx, const char *name_or_null, int json, char **out_text);
int smt_host_workspace_prune(smt_ctx *ctx, const char *name_or_null, int json, char **out_text);
/* No reference counterpart.  The workspace keeps the token ids it pooled for every stored line
 * (line_tokens.log, written by the workspace search; SEMTOOLS_TOKEN_CACHE=0 disables it).  This call re-creates every
 * stored vector from those ids with `model` -- a new embedding table behind the same tokenizer -- on the GPU, without
 * reading or tokenising the source files; document metadata is untouched.  Documents without cached tokens are listed
 * and nothing is changed.  A model with a different tokenizer is refused. */
int smt_host_workspace_reembed(smt_host_model *model, const char *name_or_null, int json, char **out_text);

void smt_host_free(char *text);

/* ---- the host layer on SEVERAL GPUs.  The caller is one process (src/bin/semtools.rs:134-135 runs the search from one
 * task); it hands the layer an smt_group instead of a context and everything above runs sharded: the embedding table is
 * replicated per GPU (smt_sharded_model), lines are pooled by all GPUs at once (smt_sharded_embed), the Documents' /
 * workspace store's matrix is an smt_sharded_corpus whose rows are dealt over the GPUs as they arrive, searches end in
 * ONE all-gather of per-shard top-k lists, the optional index is built with shared centroids.  Output bytes are the same
 * as on one GPU (the global row order is insertion order either way).  The `smt_ctx *` forms above are these with a
 * one-rank group (smt_group_from_ctx).  The group must outlive the model / calls.
 *   smt_host_group_from_spec   "0,1,2,3" = those GPUs (RCCL), "all" = every visible GPU, "<d>:<n>" = n logical shards on
 *                              GPU d (one-GPU test rigs), "<d>" = GPU d alone.  The CLI reads $SEMTOOLS_DEVICES with it.
 * The workspace store records how its rows were dealt over the GPUs (line_rows.json "shards"); the vector file itself
 * holds the rows in global order, so a store written by N GPUs opens on one, and back. */
int smt_host_group_from_spec(const char *spec, smt_group **out);
int smt_host_model_create_group(smt_group *group, const float *table, uint64_t V, int normalize, int tok_kind,
                                const char *vocab_path, const char *unk_token, smt_tokenize_cb cb, void *user,
                                uint32_t unk_id, uint32_t median_len, smt_host_model **out);
int smt_host_model_from_dir_group(smt_group *group, const char *dir, smt_host_model **out);
int smt_host_workspace_use_group(smt_group *group, const char *name, int json, char **out_text);
int smt_host_workspace_status_group(smt_group *group, const char *name_or_null, int json, char **out_text);
int smt_host_workspace_prune_group(smt_group *group, const char *name_or_null, int json, char **out_text);

/* A Hugging Face `tokenizer.json`, read natively (no Python): what model2vec-rs loads through the `tokenizers` crate
 * (call sites src/cmds/search.rs:123-128; encode_batch_fast(.., add_special_tokens = false) inside encode_with_args).
 * Supported components: BertNormalizer / Lowercase / NFD / StripAccents / Strip / Replace(String) / Sequence;
 * BertPreTokenizer / Whitespace / WhitespaceSplit / Punctuation / Metaspace / Sequence; WordPiece, Unigram; added (special) tokens.
 * Anything else fails at load time with a message naming the component.  smt_host_model_from_dir uses it when the model
 * directory holds a tokenizer.json.  encode: ids of `text` (no special tokens added); SMT_E_TRUNCATED with the true
 * count in *n_ids when cap is too small. */
typedef struct smt_host_tokenizer smt_host_tokenizer;
int smt_host_tokenizer_load(const char *tokenizer_json_path, smt_host_tokenizer **out);
void smt_host_tokenizer_free(smt_host_tokenizer *tok);
int smt_host_tokenizer_encode(smt_host_tokenizer *tok, const char *text, uint32_t *ids, uint64_t cap, uint64_t *n_ids);
int smt_host_tokenizer_info(smt_host_tokenizer *tok, uint64_t *vocab_size, int64_t *unk_id, uint64_t *median_token_bytes);
/* Wall-clock phases of this process so far as one JSON object (malloc'd): model_load, read_tokenize_embed_files,
 * embed_query, store_open_corpus_load, change_detection, embed_and_persist_changed_files, scan_select, ... -- what the
 * CLI prints to stderr under SEMTOOLS_TIMING=1. */
char *smt_host_timing_json(void);

/* Formatting primitives of the output layer, exported for tests: mode 0 = Rust `{}` of an f64,
 * 1 = Rust `{}` of an f32 (value is narrowed first), 2 = serde_json f64.  Returns malloc'd text. */
char *smt_host_format_float(double value, int mode);
/* Rust str::lines() / str::to_lowercase() as the host layer implements them: returned as "<count>\x1f<line>\x1f<line>...". */
char *smt_host_split_lines(const char *content);
char *smt_host_to_lowercase(const char *text);

#ifdef __cplusplus
}
#endif
#endif
