/*
 * semtools_oracle.h -- CPU restatement of the semtools search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under semtools_amd/ (the product) may
 * include, link, dlopen or execute this.  Allowed users: tests/, the smoke
 * check in __graft_entry__.py and the cpu_baseline leg of bench.py.
 *
 * PARITY STATUS: "parity unpinned" for the arithmetic (embed + cosine): the
 * reference (run-llama/semtools v3.0.0, pure Rust) delegates it to three
 * crates whose sources are NOT under /root/reference and cannot be built here
 * (no cargo/rustc, no network):
 *     model2vec-rs = 0.1.3   (Cargo.toml:36, Cargo.lock:2445-2448)
 *     simsimd      = 6.5.1   (Cargo.toml:37, Cargo.lock:4057-4060)
 *     qdrant-edge  = 0.0.0   (Cargo.toml:41, Cargo.lock:3081-3084)
 * and the reference's own tests pin no numeric embedding/distance value
 * (src/search/mod.rs:218-464 assert properties only).  The control flow
 * (threshold, ordering, top-k, context window, ids) IS in-tree and is pinned by
 * the reference's tests, restated in tests/test_oracle.py.
 * Each function cites the reference file:line (relative to /root/reference)
 * or the upstream crate algorithm it restates.
 */
#ifndef SEMTOOLS_ORACLE_H
#define SEMTOOLS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- A3/A4: model2vec-rs StaticModel::pool_ids (call sites
 *      src/search/mod.rs:69, src/cmds/search.rs:136,154) ------------------- */
void orc_pool_ids(const float *table, uint64_t V, uint32_t D, int normalize,
                  const uint32_t *ids, uint64_t n_ids, uint32_t max_tokens,
                  float *out);

/* CSR batch form: ids[offsets[i] .. offsets[i+1]) are line i's tokens. */
void orc_embed_lines(const float *table, uint64_t V, uint32_t D, int normalize,
                     const uint32_t *ids, const uint64_t *offsets,
                     uint64_t n_lines, uint32_t max_tokens, float *out);

/* ---- A5: simsimd 6.5.1 f32 cosine (call site src/search/mod.rs:86) ----- */
/* serial backend: f32 accumulators (SIMSIMD_MAKE_COS(serial, f32, f32, ..)) */
double orc_cosine_f32_serial(const float *a, const float *b, uint32_t n);
/* accurate backend: f64 accumulators (SIMSIMD_MAKE_COS(accurate, f32, f64, ..)) */
double orc_cosine_f32_accurate(const float *a, const float *b, uint32_t n);

/* ---- A6: search_documents (src/search/mod.rs:77-120) -------------------- */
typedef struct {
    uint64_t doc;        /* index into the documents slice                     */
    uint64_t match_line; /* 0-based line inside the doc   (mod.rs:99)          */
    uint64_t start;      /* idx.saturating_sub(n_lines)   (mod.rs:90)          */
    uint64_t end;        /* min(len, idx + n_lines + 1), exclusive (mod.rs:91) */
    double distance;     /* f64 from simsimd              (mod.rs:86)          */
} orc_result;

/* emb: all documents' line embeddings concatenated row-major [sum(lines) x D];
 * doc_line_counts[n_docs].  accurate!=0 picks the f64-accumulate cosine.
 * Returns the number of results the reference would return; writes at most
 * `cap` of them (in order) to out. */
uint64_t orc_search_documents(const float *emb, const uint64_t *doc_line_counts,
                              uint64_t n_docs, uint32_t D, const float *query,
                              uint64_t n_lines, uint64_t top_k, int has_max_distance,
                              double max_distance, int accurate, orc_result *out,
                              uint64_t cap);

/* ---- A10: Store::search_line_embeddings (src/workspace/store.rs:481-546) - */
typedef struct {
    uint32_t path_id;    /* index into the caller's path table                */
    int32_t line_number; /* 0-based (src/search/mod.rs:177)                   */
    float distance;      /* 1 - score, f32 (store.rs:531)                     */
    uint64_t row;        /* storage row, for test bookkeeping                 */
} orc_ranked_line;

/* emb [N x D] as stored (pre-normalisation happens inside, as qdrant does);
 * row_path[N], row_line[N] = payload; subset[n_subset] = path ids to keep. */
uint64_t orc_search_line_embeddings(const float *emb, const uint32_t *row_path,
                                    const int32_t *row_line, uint64_t N, uint32_t D,
                                    const float *query, const uint32_t *subset,
                                    uint64_t n_subset, uint64_t top_k,
                                    int has_max_distance, float max_distance,
                                    orc_ranked_line *out, uint64_t cap);

/* ---- A9: ids (src/workspace/store.rs:75-89, 651-661) -------------------- */
uint64_t orc_fnv1a_hash(const uint8_t *bytes, uint64_t n);
uint64_t orc_line_embedding_id(const char *path, int32_t line_number);
uint64_t orc_doc_meta_id(const char *path);

/* ---- CPU baselines for bench.py (cpu_baseline.kind = "port") ----------- */
/* Reference-faithful A6 over one flat corpus is orc_search_documents with
 * n_docs=1.  The "fair" variant below is NOT a restatement: threaded scan,
 * per-thread bounded lists, same distances (f32 serial per row). */
uint64_t orc_scan_topk_threads(const float *emb, uint64_t N, uint32_t D,
                               const float *query, uint64_t top_k, int n_threads,
                               uint64_t *out_rows, double *out_dist);

/* bench.py's honest single-core baseline: the reference's control flow with a SIMD (runtime-dispatched AVX-512 /
 * AVX2+FMA) f32 cosine, see cpu_fast.c */
uint64_t orc_search_documents_simd(const float *emb, uint64_t N, uint32_t D, const float *query, uint64_t n_lines,
                                   uint64_t top_k, int has_max_distance, double max_distance, orc_result *out, uint64_t cap);
const char *orc_simd_backend(void);

#ifdef __cplusplus
}
#endif
#endif
