/*
 * semtools_hip.h -- C ABI of libsemtools_hip.so, the MI355X (gfx950) core for
 * the `semtools search` / `semtools workspace` hot path:
 *
 *     embed (token-id gather + mean-pool + L2-normalise)
 *       -> cosine scan of query vector(s) over the row-major f32 corpus
 *       -> top-k / max-distance selection
 *
 * The reference (run-llama/semtools v3.0.0, /root/reference) has NO FFI seam
 * for this path: the arithmetic sits behind three Rust crates (model2vec-rs,
 * simsimd, qdrant-edge) called from ordinary Rust functions.  Each entry point
 * below names the reference interface (file:line, relative to /root/reference)
 * it replaces; INTEGRATION.md shows the Rust `extern "C"` stub a maintainer
 * would add at those call sites.
 *
 * Conventions
 *   - Plain C: opaque handles, raw pointers, sizes.  No torch/C++ types.
 *   - Every function returning `int` returns SMT_OK (0) or a negative SMT_E_*;
 *     smt_last_error() returns a thread-local message for the last failure.
 *     Nothing unwinds or aborts across the ABI.
 *   - Host buffers are owned by the caller; device memory lives behind handles.
 *   - Handles are not thread-safe: serialise calls per handle (the reference
 *     calls this path synchronously from one task, src/bin/semtools.rs:134).
 *   - One smt_ctx == one GPU + one HIP stream.  Multi-GPU lives behind smt_group /
 *     smt_sharded_corpus (bottom of this file): the library owns one context per GPU and
 *     the RCCL communicator; a single host thread drives the whole node, or one rank
 *     per process joins the same communicator (smt_group_create_rank).
 *   - "_device" entry points take/return DEVICE pointers and enqueue on the
 *     context's stream without synchronising (bench / multi-GPU pipelines).
 *   - There is no CPU fallback: every compute entry point fails with
 *     SMT_E_HIP when no gfx950 device is usable.
 *
 * Domain (semtools_amd/csrc/domain.hip)
 *   A row or query vector is IN DOMAIN iff every component is finite and its largest magnitude is 0 or lies in
 *   [2^-40, 2^40].  Inside it the returned distance is simsimd's cosine in its f64 "accurate" form, bit for bit, and
 *   scaling a vector by a power of two changes no answer.  Outside it the reference itself has no single answer --
 *   cos_finish turns a NaN into distance 0.0 (the BEST score: src/search/mod.rs:86-89 pushes it, :107-111 sorts it
 *   first) and the f32 accumulators of simsimd's serial / SIMD backends overflow beyond |x| ~ 1.8e19 and underflow
 *   below ~1e-19 in an order-dependent way -- and the f32 / fp16 nominating kernels and their error bounds do not
 *   cover it.  Such vectors are REFUSED at the boundary instead of being answered for:
 *     - rows, where they enter a corpus (smt_corpus_append_host, smt_corpus_write_rows, smt_corpus_from_device,
 *       smt_embed with append_to, the sharded forms of these): SMT_E_INVALID, the corpus unchanged, the message
 *       names the first offending row; the file loaders return SMT_E_IO (a damaged file);
 *     - queries of the host-form searches: SMT_E_INVALID before anything is launched;
 *     - queries of the *_device entry points: SMT_STATUS_INVALID_QUERY in the _ex form's status word (and the
 *       context's uncertain counter); that query's list means nothing.
 *   model2vec's pool step with `normalize` emits unit or zero rows: nothing the reference's own path produces is
 *   ever refused.  A corpus adopted with smt_corpus_from_device is checked once, as it is at that call.
 */
#ifndef SEMTOOLS_HIP_H
#define SEMTOOLS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMT_OK 0
#define SMT_E_INVALID (-1)     /* bad argument                                   */
#define SMT_E_HIP (-2)         /* HIP runtime / device error, or no GPU          */
#define SMT_E_NOMEM (-3)       /* host or device allocation failed / capacity    */
#define SMT_E_TRUNCATED (-4)   /* more hits than out_cap; counts hold true sizes */
#define SMT_E_IO (-5)          /* file error in save/load                        */
#define SMT_E_UNSUPPORTED (-6) /* e.g. embedding dim other than 256              */

#define SMT_DIM 256u /* LINE_EMBEDDING_SIZE, src/workspace/store.rs:37 */

/* selection semantics of smt_search */
#define SMT_MODE_DOCUMENTS 0 /* search_documents, src/search/mod.rs:77-120            */
#define SMT_MODE_WORKSPACE 1 /* Store::search_line_embeddings, store.rs:481-546        */

typedef struct smt_ctx smt_ctx;
typedef struct smt_model smt_model;
typedef struct smt_corpus smt_corpus;

/* half-open range of corpus rows [begin, end) */
typedef struct smt_range {
    uint64_t begin;
    uint64_t end;
} smt_range;

/* ------------------------------------------------------------------ context */

/* Bind to GPU `device` and create a private (non-blocking) HIP stream.  Everything the library enqueues is ordered on THAT stream
 * only: device memory handed in (smt_corpus_from_device, the *_device entry points) must be complete before the call -- synchronise
 * the stream that produced it, or create the context on that stream (below). */
int smt_ctx_create(int device, smt_ctx **out);
/* Same, but enqueue on an existing hipStream_t owned by the caller (e.g. the
 * host framework's current stream).  NULL here means THE NULL STREAM. */
int smt_ctx_create_on_stream(int device, void *stream, smt_ctx **out);
void smt_ctx_destroy(smt_ctx *ctx);
int smt_ctx_synchronize(smt_ctx *ctx);
const char *smt_last_error(void);
const char *smt_version(void);
/* number of visible HIP devices, or SMT_E_HIP */
int smt_device_count(void);

/* Per-kernel timing with HIP events recorded on the context's stream around
 * each launch of the named kernel family ("scan", "gemm", "embed", "select").
 * Reading synchronises the stream. */
int smt_prof_enable(smt_ctx *ctx, int on);
int smt_prof_reset(smt_ctx *ctx);
int smt_prof_read(smt_ctx *ctx, const char *kernel, uint64_t *launches, double *total_ms);

/* -------------------------------------------------------------------- model
 * Replaces the device half of StaticModel::from_pretrained (call sites
 * src/cmds/search.rs:123-128, src/cmds/ask.rs:128-133): the f32 `embeddings`
 * table [V x D] is uploaded once and stays resident.  `normalize` is the
 * model's config flag (true for potion-multilingual-128M). */
int smt_model_create(smt_ctx *ctx, const float *table_host, uint64_t V, uint32_t D,
                     int normalize, smt_model **out);
/* Stream an f32 table [V x D] that sits at `byte_offset` of a file (the `embeddings` tensor of model.safetensors)
 * straight into HBM through two pinned buffers: the file read of chunk j+1 overlaps the PCIe copy of chunk j, and no
 * pageable staging copy of the whole table (512 MB for potion-multilingual-128M) is ever made. */
int smt_model_create_from_file(smt_ctx *ctx, const char *path, uint64_t byte_offset, uint64_t V, uint32_t D,
                               int normalize, smt_model **out);
/* adopt a table already in device memory (not copied, not freed) */
int smt_model_create_from_device(smt_ctx *ctx, const float *table_dev, uint64_t V, uint32_t D,
                                 int normalize, smt_model **out);
void smt_model_destroy(smt_model *model);

/* Replaces the pool step of StaticModel::encode_with_args / encode_single
 * (call sites src/search/mod.rs:69,138,153; src/cmds/search.rs:136,154).
 * Tokenisation stays on the host: `ids` are the unk-filtered token ids of all
 * lines back to back, `offsets[n_lines+1]` the CSR boundaries.  Each line is
 * truncated to `max_tokens` ids (2048 for lines, 512 for queries; 0 = no cap),
 * rows are summed in token order in f32, divided by the count, and (if the
 * model normalises) divided by max(||v||, 1e-12).  An empty line gives the
 * zero vector.  Output goes to `out_host` [n_lines x D] and/or is appended to
 * `append_to` (first new row index in *first_row). Either may be NULL. */
int smt_embed(smt_model *model, const uint32_t *ids, const uint64_t *offsets, uint64_t n_lines,
              uint32_t max_tokens, float *out_host, smt_corpus *append_to, uint64_t *first_row);
/* everything resident: ids/offsets/out are device pointers; async on the stream */
int smt_embed_device(smt_model *model, const uint32_t *ids_dev, const uint64_t *offsets_dev,
                     uint64_t n_lines, uint32_t max_tokens, float *out_dev);

/* ------------------------------------------------------------------- corpus
 * Replaces `Document::embeddings: Vec<Vec<f32>>` (src/search/mod.rs:18-22)
 * and the vector half of the workspace store (line_embeddings.qdrant,
 * src/workspace/store.rs:152-166, upsert :402-434): one row-major f32 matrix
 * [rows x D] resident in HBM; row index == insertion order. */
int smt_corpus_create(smt_ctx *ctx, uint32_t D, uint64_t capacity_rows, smt_corpus **out);
/* adopt rows already in device memory (not copied, not freed, not growable) */
int smt_corpus_from_device(smt_ctx *ctx, const float *rows_dev, uint64_t n_rows, uint32_t D,
                           smt_corpus **out);
void smt_corpus_destroy(smt_corpus *corpus);
int smt_corpus_append_host(smt_corpus *corpus, const float *rows, uint64_t n_rows,
                           uint64_t *first_row);
/* overwrite existing rows [first_row, first_row+n_rows) (upsert of a changed doc) */
int smt_corpus_write_rows(smt_corpus *corpus, uint64_t first_row, const float *rows,
                          uint64_t n_rows);
int smt_corpus_read_rows(smt_corpus *corpus, uint64_t first_row, uint64_t n_rows, float *out_host);
int smt_corpus_truncate(smt_corpus *corpus, uint64_t n_rows);
/* The fp16 OPERAND IMAGE of a corpus: what the batched nomination modes multiply with (unit rows x 2^10 as fp16 in MFMA operand
 * order, 512 B per row beside the 1 KiB of f32), so that a batch of >= 8 queries reads half the bytes per row and converts
 * nothing.  Derived data only: nominations come from it, every returned distance is re-scored from the f32 rows, results are
 * identical with and without it.  A corpus whose memory the library owns builds and maintains it by itself (first batch of
 * >= 8 queries over >= 64 Ki rows, or the fourth smaller search of a shard of >= 4 M rows, which then answers single queries
 * from it as well; appends, writes and truncation are tracked; tuning key corpus_image = 0 turns that off).
 * smt_corpus_prepack(corpus, 1) builds it NOW from the rows as they are -- the way to have one for a corpus adopted with
 * smt_corpus_from_device, whose caller then answers for calling it again after changing rows; (corpus, 0) drops it and keeps the
 * corpus without one.  The shards of an smt_sharded_corpus are corpora like any other (smt_sharded_corpus_shard hands them out):
 * shards the library filled keep their images by themselves, adopted shards get one per shard through this call.
 * No reference counterpart (the reference scores Vec<Vec<f32>> rows one by one, src/search/mod.rs:84-119). */
int smt_corpus_prepack(smt_corpus *corpus, int enable);
uint64_t smt_corpus_image_bytes(const smt_corpus *corpus);
uint64_t smt_corpus_rows(const smt_corpus *corpus);
uint32_t smt_corpus_dim(const smt_corpus *corpus);
/* flat little-endian file: 32-byte header + rows*D f32 */
int smt_corpus_save(smt_corpus *corpus, const char *path);
int smt_corpus_load(smt_ctx *ctx, const char *path, smt_corpus **out);
/* Incremental flush: `path` must already hold exactly the first `rows_on_disk` rows of this corpus;
 * rows [rows_on_disk, rows) are appended and the header is updated (O(new rows), not O(corpus)). */
int smt_corpus_append_to_file(smt_corpus *corpus, const char *path, uint64_t rows_on_disk);

/* ------------------------------------------------------------------- search
 * Replaces f32::cosine + the selection in search_documents
 * (src/search/mod.rs:84-119) and Store::search_line_embeddings
 * (src/workspace/store.rs:481-546).
 *
 *   queries     host, [nq x D] f32
 *   max_distance NaN = "None".
 *   mode SMT_MODE_DOCUMENTS: max_distance None -> the top_k rows by
 *        (distance asc, row asc) [= the reference's stable sort + take(top_k)];
 *        max_distance given -> ALL rows with distance < max_distance (strict),
 *        same order, top_k ignored.
 *   mode SMT_MODE_WORKSPACE: rows with score > 1 - max_distance (f32, if given),
 *        then ALWAYS the top_k of them; top_k == 0 -> nothing.
 *   ranges      optional filter: only rows inside these sorted, disjoint
 *        ranges are scanned (workspace path subset; a document's lines are
 *        contiguous rows).  NULL/0 = whole corpus.
 *   row_base    added to every returned row (global index of this shard's row 0).
 *   out_rows/out_dist  [nq x out_cap]; out_counts[nq] = hits for each query
 *        (the TRUE count even when > out_cap, in which case the call returns
 *        SMT_E_TRUNCATED after filling the first out_cap of each).
 *
 * Distances are f64: the f32 scan only nominates candidates; every returned
 * distance is recomputed on the GPU with f64 accumulation in index order --
 * simsimd's "accurate" formula -- so results do not depend on reduction order.
 */
int smt_search(smt_corpus *corpus, const float *queries, uint32_t nq, uint32_t top_k,
               double max_distance, int mode, const smt_range *ranges, uint32_t n_ranges,
               uint64_t row_base, uint64_t *out_rows, double *out_dist, uint64_t *out_counts,
               uint64_t out_cap);

/* Resident top-k (mode DOCUMENTS, no threshold, no ranges): queries_dev
 * [nq x D] and outputs [nq x top_k] are device pointers; unused slots are
 * (row = UINT64_MAX, dist = +inf).  Enqueues on the stream, no sync.  1 <= top_k <= 56. */
int smt_search_topk_device(smt_corpus *corpus, const float *queries_dev, uint32_t nq,
                           uint32_t top_k, uint64_t row_base, uint64_t *out_rows_dev,
                           double *out_dist_dev);
/* The same with a verdict PER QUERY: search_documents returns a definite list (src/search/mod.rs:107-119), so a caller that
 * pipelines device calls must be able to tell which answer of which call is the proved exact top-k and which is not.
 * out_status_dev [nq] (device-addressable: HBM or pinned host; NULL = smt_search_topk_device) receives, in stream order with
 * the lists:
 *   SMT_STATUS_PROVED     the list IS the exact top-k (the select's exactness certificate held: "Exactness bookkeeping" below)
 *   SMT_STATUS_UNCERTAIN  more near-ties around the k-th place than the guard band holds: every returned (row, distance) pair is
 *                         exact, but a row outside the list may belong to it -- re-ask this query through smt_search
 *   SMT_STATUS_OVERFLOW   batched path only: a candidate buffer overflowed (adversarial row order) and rows were dropped on the
 *                         way; same remedy
 *   SMT_STATUS_INVALID_QUERY  the query is outside the library's domain ("Domain" at the top of this file: a non-finite
 *                         component or an absurd magnitude); the list means nothing.  smt_search refuses such a query instead
 * Queries with a non-zero status are also counted by smt_ctx_uncertain_count.  smt_search / smt_sharded_search never return such
 * a list: they re-answer the query exhaustively themselves. */
#define SMT_STATUS_PROVED 0
#define SMT_STATUS_UNCERTAIN 1
#define SMT_STATUS_OVERFLOW 2
#define SMT_STATUS_INVALID_QUERY 3
int smt_search_topk_device_ex(smt_corpus *corpus, const float *queries_dev, uint32_t nq,
                              uint32_t top_k, uint64_t row_base, uint64_t *out_rows_dev,
                              double *out_dist_dev, uint32_t *out_status_dev);

/* Merge `n_lists` sorted (dist asc, row asc) lists of length k_in each (padded
 * with UINT64_MAX/+inf) per query into the best k_out.  Inputs are laid out
 * [n_lists][nq][k_in] -- exactly what an all-gather of per-rank
 * smt_search_topk_device outputs produces.  Host version: */
int smt_merge_topk(const uint64_t *rows, const double *dist, uint32_t n_lists, uint32_t nq,
                   uint32_t k_in, uint32_t k_out, uint64_t *out_rows, double *out_dist,
                   uint64_t *out_counts);
/* device version (all pointers device, async on the stream) */
int smt_merge_topk_device(smt_ctx *ctx, const uint64_t *rows_dev, const double *dist_dev,
                          uint32_t n_lists, uint32_t nq, uint32_t k_in, uint32_t k_out,
                          uint64_t *out_rows_dev, double *out_dist_dev);

/* Packed variant: one buffer per rank [nq][2][k] of 8-byte words -- row ids, then the f64
 * distance bits -- so that ONE all-gather moves both.  packed_dev is the gathered
 * [n_lists][nq][2][k_in]; out_packed_dev receives [nq][2][k_out].  Device pointers, async. */
int smt_merge_topk_packed_device(smt_ctx *ctx, const uint64_t *packed_dev, uint32_t n_lists, uint32_t nq,
                                 uint32_t k_in, uint32_t k_out, uint64_t *out_packed_dev);

/* ----------------------------------------------------------------- IVF-PQ
 * Approximate index over a resident corpus (BASELINE config 5).  There is NO reference
 * counterpart: the reference's workspace store scans exactly (README's "IVF_PQ" is a stale
 * label); the contract is recall against smt_search on the same corpus.  Every returned
 * (row, distance) is exact -- candidates from the ADC scan are re-ranked with the exact f64
 * distance -- only top-k membership is approximate.  The index keeps a pointer to `corpus`,
 * which must stay alive and unchanged.  It is built for what model2vec emits -- UNIT rows (zero
 * rows are fine): its quantisers work on the rows as they are, not on their directions, so
 * smt_ivfpq_build / smt_ivfpq_append refuse a corpus holding rows of other lengths
 * (| |x|^2 - 1 | > 1e-3) with SMT_E_UNSUPPORTED; the exact searches have no such condition. */
typedef struct smt_ivfpq smt_ivfpq;
typedef struct smt_ivfpq_params {
    uint32_t nlist;        /* coarse lists: multiple of 32 in [32, 4096]                  */
    uint32_t m;            /* PQ sub-quantisers: 32 (dsub = 8)                            */
    uint32_t nbits;        /* 8                                                           */
    uint32_t train_iters;  /* k-means iterations for both quantisers (0 = 10)            */
    uint64_t train_sample; /* training rows, evenly spaced (0 = 64 * nlist)              */
    uint32_t reserved;     /* must be 0 (was `refine`: an int8 refinement stage, measured +6 % for a 7x larger index, removed) */
    uint32_t local_pca;    /* 0 = one global residual codebook set (8 dims x 256 codes per sub-quantiser);    */
                           /* 1 = per-list PCA: every list gets its own orthonormal basis of its 32 principal */
                           /* residual directions and one 8-bit scalar quantiser per direction (still 32 B    */
                           /* per row; + 32 KiB per list).  Ranks the rows inside a list far better on        */
                           /* clustered data -- the workspace store builds its index this way                 */
} smt_ivfpq_params;
int smt_ivfpq_build(smt_corpus *corpus, const smt_ivfpq_params *params, smt_ivfpq **out);
void smt_ivfpq_destroy(smt_ivfpq *index);
/* nprobe lists scanned per query; in every probed list the `rerank` best ADC candidates
 * (0 = 512; range [4, 512]) are re-scored against the full-precision rows inside the scan
 * kernel, and the best top_k + 8 of all lists get the exact f64 distance.  top_k <= 56.
 * Outputs as in smt_search. */
int smt_ivfpq_search(smt_ivfpq *index, const float *queries, uint32_t nq, uint32_t top_k, uint32_t nprobe,
                     uint32_t rerank, uint64_t row_base, uint64_t *out_rows, double *out_dist,
                     uint64_t *out_counts, uint64_t out_cap);
/* Device-resident form: queries [nq x 256] f32 and the outputs [nq x top_k] (rows u64, padded with
 * UINT64_MAX; exact f64 distances, padded with +inf) live in device (or pinned host) memory; enqueued on
 * the context's stream, complete after smt_ctx_synchronize. */
int smt_ivfpq_search_device(smt_ivfpq *index, const float *queries_dev, uint32_t nq, uint32_t top_k, uint32_t nprobe,
                            uint32_t rerank, uint64_t row_base, uint64_t *out_rows_dev, double *out_dist_dev);
/* build_ms4 = {coarse k-means, assign all rows, PQ training, sort + encode} */
int smt_ivfpq_info(const smt_ivfpq *index, uint64_t *n_rows, uint32_t *nlist, uint64_t *index_bytes,
                   double *build_ms4);
int smt_ivfpq_list_sizes(const smt_ivfpq *index, uint64_t *sizes_host /* [nlist] */);
/* Persist / restore the index (centroids, quantisers, list table, ids, codes: ~36 B per row).  The file
 * refers to corpus rows by position: load fails with SMT_E_INVALID unless `corpus` holds AT LEAST the
 * row count the index covers (then smt_ivfpq_append if the corpus only grew, else rebuild -- 0.5 s per 10 M rows). */
int smt_ivfpq_save(smt_ivfpq *index, const char *path);
int smt_ivfpq_load(smt_corpus *corpus, const char *path, smt_ivfpq **out);
/* Incremental insert: rows appended to the corpus since the index was built (or last extended) are assigned to
 * their nearest list and encoded with that list's EXISTING quantiser (no retraining), then merged into the
 * inverted lists -- O(new rows) arithmetic plus one 36 B/row re-layout.  *n_added = rows taken in (may be NULL).
 * The quantisers drift away from the data as it grows: rebuild when the corpus has roughly doubled. */
int smt_ivfpq_append(smt_ivfpq *index, uint64_t *n_added);

/* ------------------------------------------------------- groups of GPUs (RCCL)
 * The reference runs the whole search synchronously from ONE task of ONE process (src/bin/semtools.rs:134-135,
 * src/cmds/search.rs:197-206); these entry points let that one caller use every GPU of the node.  There is no
 * reference counterpart for the sharding itself; the contract is: sharded result == single-shard result.
 *
 *   smt_init(devices, n)            process-wide default group; NULL / 0 = every visible GPU;
 *                                   idempotent for the same list.  smt_shutdown() releases it.
 *   smt_group_create(devices, n)    single process: one context + stream per device, ncclCommInitAll.
 *   smt_group_unique_id / smt_group_create_rank
 *                                   one rank per process (torchrun / MPI): rank 0 makes the 128-byte id, the host
 *                                   broadcasts it by any means, every process joins with its device and rank
 *                                   (ncclCommInitRank).  Calls on the group are then SPMD: every process makes
 *                                   the same calls with the same host arguments.
 * RCCL (librccl.so.1) is loaded when the first group is created, not when the library is. */
typedef struct smt_group smt_group;
typedef struct smt_sharded_corpus smt_sharded_corpus;
#define SMT_UNIQUE_ID_BYTES 128
int smt_init(const int *devices, int n_dev);
int smt_shutdown(void);
smt_group *smt_default_group(void);
int smt_group_create(const int *devices, int n_dev, smt_group **out);
/* n_shards logical ranks on ONE device, exchanging through device copies instead of RCCL (RCCL refuses two
 * ranks on one GPU): lets a single-GPU box run the whole sharded path -- the GPU tests do -- and splits a
 * corpus into independently growable shards.  smt_group_info reports rccl_ranks = 0 for such a group. */
int smt_group_create_logical(int device, int n_shards, smt_group **out);
/* A ONE-rank group around an existing context (which the caller keeps and destroys after the group): every smt_sharded_*
 * entry point then forwards to its single-GPU counterpart on that context -- no RCCL is loaded, nothing is exchanged.
 * It lets a host layer written against the sharded API serve the default single-GPU case at single-GPU cost. */
int smt_group_from_ctx(smt_ctx *ctx, smt_group **out);
int smt_group_unique_id(void *id_out /* SMT_UNIQUE_ID_BYTES */);
int smt_group_create_rank(int device, int rank, int n_ranks, const void *unique_id, smt_group **out);
void smt_group_destroy(smt_group *group);
/* n_ranks: size of the group; n_local: GPUs driven by this process (ranks first_rank .. first_rank+n_local-1);
 * rccl_ranks: what ncclCommCount reports for the communicator; rccl_version: ncclGetVersion.  Any may be NULL. */
int smt_group_info(const smt_group *group, int *n_ranks, int *n_local, int *first_rank, int *rccl_ranks, int *rccl_version);
/* the context of local device i (tuning keys, profiling, smt_model_create / smt_embed on that GPU); NULL if out of range */
smt_ctx *smt_group_ctx(smt_group *group, int local_index);
int smt_group_synchronize(smt_group *group); /* every local stream, async pipelines drained */
int smt_group_barrier(smt_group *group);     /* + an all-gather across the ranks */
/* How the per-shard k-lists of a sharded top-k search meet ("all-gather of the per-shard top-k"):
 *   SMT_TRANSPORT_RCCL  one ncclAllGather of the packed lists, then merge_topk_kernel.  The only choice when the ranks are
 *                       separate processes (smt_group_create_rank); available to smt_group_create groups.
 *   SMT_TRANSPORT_COPY  logical groups: the same all-gather made of event-ordered device copies.
 *   SMT_TRANSPORT_PEER  one-process groups (smt_group_create when every device can read every other's memory -- peer access is
 *                       enabled at creation --, smt_group_create_logical): nothing is gathered; the merge kernel of the device
 *                       that needs the answer reads the ranks' lists where they lie, ordered by one stream event per rank.
 *                       DEFAULT wherever it is possible: an answer costs n - 1 stream waits (enqueued by the ranks' issuing threads)
 *                       and one launch instead of RCCL's per-call host cost for a 960-byte payload.
 * $SEMTOOLS_GROUP_TRANSPORT = rccl | copy | peer overrides the default at creation (ignored where impossible).
 * smt_group_set_transport synchronises the group first; SMT_E_INVALID if the group cannot use that transport.
 * Threshold mode / large k (host-list exchange), barriers and the shared-centroid all-reduce always use RCCL (or copies). */
#define SMT_TRANSPORT_RCCL 0
#define SMT_TRANSPORT_COPY 1
#define SMT_TRANSPORT_PEER 2
int smt_group_set_transport(smt_group *group, int transport);
int smt_group_transport(const smt_group *group); /* SMT_TRANSPORT_*; negative = error */

/* A corpus row-sharded over the group.  Global row = position in the whole corpus = INSERTION ORDER (the reference's tie
 * order: document, then line -- src/search/mod.rs:84-85,107-111).  A corpus made in one go is cut into contiguous ranges,
 * rows_per_rank = ceil(N / n_ranks) (a document's lines and the path-subset ranges stay ranges).  A corpus
 * that GROWS (the workspace store: src/workspace/store.rs:402-434 appends, never rewrites) deals every append over the
 * ranks -- the emptier shards are filled first, a handful of rows goes to one shard -- so that all GPUs keep equal shares;
 * the numbering is then a list of pieces (smt_sharded_corpus_layout).  Inside a shard local order == global order.
 *   create       empty corpus (rows arrive through smt_sharded_embed / append_host)
 *   from_host    `rows` is the WHOLE matrix [total_rows x D]; every process uploads the slices of its local ranks.
 *   from_device  adopt one resident buffer per LOCAL device (not copied, not freed); the ranks' sizes are
 *                exchanged with one all-gather.
 *   load / save / append_to_file   the smt_corpus_save file format (rows in global order -- a file written by N GPUs
 *                loads on one, and back); every rank streams its own pieces.  load cuts the file into ceil(N / n_ranks)
 *                ranges; load_layout restores a recorded piece list (piece k = piece_rows[k] consecutive global rows on
 *                rank piece_rank[k]), e.g. so that per-shard index files stay valid.  A failing rank reports through the
 *                collective: every process returns the error.
 *   layout       the piece list in global order; returns the piece count (call with cap 0 to size the arrays).
 *   append_host  n_rows new global rows, dealt to the ranks as described above; every process passes the same rows.
 *   read_rows / write_rows   rows by global position <-> host (single-process groups: the one caller sees the matrix).
 *   shard        the smt_corpus behind local device i and its row count; row_base (may be NULL) is only defined -- and
 *                only accepted -- while the corpus is one range per rank.  Do not append to a shard directly. */
int smt_sharded_corpus_create(smt_group *group, uint32_t D, smt_sharded_corpus **out);
int smt_sharded_corpus_from_host(smt_group *group, const float *rows, uint64_t total_rows, uint32_t D, smt_sharded_corpus **out);
int smt_sharded_corpus_from_device(smt_group *group, const float *const *shard_rows_dev, const uint64_t *shard_rows,
                                   uint32_t D, smt_sharded_corpus **out);
int smt_sharded_corpus_load(smt_group *group, const char *path, smt_sharded_corpus **out);
int smt_sharded_corpus_load_layout(smt_group *group, const char *path, const uint64_t *piece_rows, const uint32_t *piece_rank,
                                   uint64_t n_pieces, smt_sharded_corpus **out);
uint64_t smt_sharded_corpus_layout(const smt_sharded_corpus *corpus, uint64_t *piece_rows, uint32_t *piece_rank, uint64_t cap);
int smt_sharded_corpus_save(smt_sharded_corpus *corpus, const char *path);
int smt_sharded_corpus_append_to_file(smt_sharded_corpus *corpus, const char *path, uint64_t rows_on_disk);
/* The same in two steps, so that persisting overlaps embedding (src/workspace/store.rs:402-434 flushes every 1000-point chunk while
 * it goes; a 1 M-line workspace is 1 GB of rows whose fsync takes as long as tokenising + pooling them):
 *   SMT_APPEND_WRITE_AHEAD  rows [rows_written, rows) go to their places in the file and the kernel is asked to start writing them
 *                           out (sync_file_range), but nothing is durable and the header still names rows_on_disk rows -- a crash
 *                           leaves the old, consistent prefix.  Call it after every embedded batch; one-shard corpora only
 *                           (SMT_E_UNSUPPORTED otherwise: keep the rows for the commit).
 *   SMT_APPEND_CREATE       rows_on_disk == 0 and no file yet: start one (empty header).
 *   flags 0                 the commit: rows [rows_written, rows) written, fsync, header last.  rows_on_disk <= rows_written: what
 *                           earlier WRITE_AHEAD calls put there.  smt_sharded_corpus_append_to_file(c, p, r) == _ex(c, p, r, r, 0). */
#define SMT_APPEND_WRITE_AHEAD 1
#define SMT_APPEND_CREATE 2
int smt_sharded_corpus_append_to_file_ex(smt_sharded_corpus *corpus, const char *path, uint64_t rows_on_disk, uint64_t rows_written,
                                         int flags);
void smt_sharded_corpus_destroy(smt_sharded_corpus *corpus);
uint64_t smt_sharded_corpus_rows(const smt_sharded_corpus *corpus);
int smt_sharded_corpus_rank_rows(const smt_sharded_corpus *corpus, uint64_t *rows_per_rank /* [n_ranks] */);
int smt_sharded_corpus_shard(smt_sharded_corpus *corpus, int local_index, smt_corpus **shard, uint64_t *row_base, uint64_t *rows);
int smt_sharded_corpus_append_host(smt_sharded_corpus *corpus, const float *rows, uint64_t n_rows, uint64_t *first_row);
int smt_sharded_corpus_read_rows(smt_sharded_corpus *corpus, uint64_t first_row, uint64_t n_rows, float *out_host);
int smt_sharded_corpus_write_rows(smt_sharded_corpus *corpus, uint64_t first_row, const float *rows, uint64_t n_rows);

/* The embedding table replicated on every device of the group, and K1 sharded by line ("K1 shards by
 * line with a replicated table", no collective): the device half of StaticModel::from_pretrained / encode_with_args
 * (call sites src/search/mod.rs:69,138,153; src/cmds/search.rs:123-128,136,154) for a caller that owns N GPUs.
 * smt_sharded_embed = smt_embed's arguments and results: the lines are dealt to the ranks in contiguous blocks, block r is
 * pooled on rank r from its copy of the table (one host thread per device) and, with `append_to`, appended to rank r's
 * shard; the blocks in rank order become the new global rows, so global row == line order and *first_row = the first of
 * them.  Every row is bit-identical to smt_embed's (same kernel, same table).  out_host receives [n_lines x D] (in a
 * multi-process group: the blocks of the local ranks only).  If any rank fails, no shard keeps rows of the call. */
typedef struct smt_sharded_model smt_sharded_model;
int smt_sharded_model_create(smt_group *group, const float *table_host, uint64_t V, uint32_t D, int normalize, smt_sharded_model **out);
int smt_sharded_model_create_from_file(smt_group *group, const char *path, uint64_t byte_offset, uint64_t V, uint32_t D,
                                       int normalize, smt_sharded_model **out);
void smt_sharded_model_destroy(smt_sharded_model *model);
int smt_sharded_embed(smt_sharded_model *model, const uint32_t *ids, const uint64_t *offsets, uint64_t n_lines, uint32_t max_tokens,
                      float *out_host, smt_sharded_corpus *append_to, uint64_t *first_row);

/* smt_search over the sharded corpus: same arguments and semantics (ranges and returned rows are GLOBAL), same
 * result as smt_search on the unsharded matrix.  top_k <= 56 without "all under threshold": per-device scan +
 * select -> ONE ncclAllGather of the packed [nq][2][k] lists -> merge_topk_kernel.  Threshold mode and larger k:
 * all-gather of the hit counts, one all-gather of a max-count-padded buffer, host merge. */
int smt_sharded_search(smt_sharded_corpus *corpus, const float *queries, uint32_t nq, uint32_t top_k, double max_distance,
                       int mode, const smt_range *ranges, uint32_t n_ranges, uint64_t *out_rows, double *out_dist,
                       uint64_t *out_counts, uint64_t out_cap);
/* Device-resident top-k (mode DOCUMENTS, no threshold, no ranges), nothing synchronises: queries_dev[i] is a
 * pointer ON LOCAL DEVICE i to the nq queries; out_packed[i] (NULL = this device does not need the answer) is
 * device-addressable memory of local device i -- HBM or pinned host -- receiving [nq][2][top_k] 8-byte words:
 * global rows (padding UINT64_MAX), then the f64 distance bit patterns (padding +inf).  With tuning key
 * async_select set on the contexts and nq == 1, the select, the all-gather and the merge of call i run on the
 * contexts' aux streams while the scan of call i+1 streams. */
int smt_sharded_search_topk_device(smt_sharded_corpus *corpus, const float *const *queries_dev, uint32_t nq, uint32_t top_k,
                                   uint64_t *const *out_packed);
/* ... with the per-query verdict of smt_search_topk_device_ex: out_status (NULL = none) holds one pointer per LOCAL device,
 * out_status[i] (NULL = not wanted there; needs out_packed[i]) receiving [nq] SMT_STATUS_* codes -- the WORST status any shard
 * of the group reported for that query (every rank's status words travel with its k-lists: same all-gather, or read in place by
 * the peer transport). */
int smt_sharded_search_topk_device_ex(smt_sharded_corpus *corpus, const float *const *queries_dev, uint32_t nq, uint32_t top_k,
                                      uint64_t *const *out_packed, uint32_t *const *out_status);

/* IVF index over a sharded corpus: every rank indexes ITS rows.  shared_centroids != 0 makes the build data-parallel
 * (SURVEY.md 8(e)): the coarse k-means runs on every rank's sample of its own rows and the fixed-point centroid
 * sums + counts are summed over the ranks after every accumulation (ncclAllReduce on the ranks' streams), so all
 * ranks end with the SAME nlist centroids -- "nlist lists over the whole corpus", each list spread over the shards --
 * while quantisers and codes are fitted locally.  shared_centroids == 0: independent per-shard indexes (own centroids,
 * no collective in the build).  Search: per-shard smt_ivfpq_search with global rows -> the same all-gather + merge as
 * smt_sharded_search.  Exact distances, approximate membership; top_k <= 56. */
typedef struct smt_sharded_ivfpq smt_sharded_ivfpq;
int smt_sharded_ivfpq_build(smt_sharded_corpus *corpus, const smt_ivfpq_params *params, int shared_centroids,
                            smt_sharded_ivfpq **out);
void smt_sharded_ivfpq_destroy(smt_sharded_ivfpq *index);
smt_ivfpq *smt_sharded_ivfpq_shard(smt_sharded_ivfpq *index, int local_index); /* smt_ivfpq_info etc.; NULL if out of range */
int smt_sharded_ivfpq_search(smt_sharded_ivfpq *index, const float *queries, uint32_t nq, uint32_t top_k, uint32_t nprobe,
                             uint32_t rerank, uint64_t *out_rows, double *out_dist, uint64_t *out_counts, uint64_t out_cap);
/* Life cycle, shard by shard (smt_ivfpq_save / _load / _append / _info on every local rank).  A one-rank group uses `path`
 * itself; otherwise rank r's part is `<path>.r<r>of<n_ranks>` -- the files name LOCAL rows by position, so they are valid
 * for the layout they were built on (persist it with smt_sharded_corpus_layout, restore it with _load_layout).
 * info: rows covered and index bytes summed over the local ranks, the largest nlist. */
int smt_sharded_ivfpq_save(smt_sharded_ivfpq *index, const char *path);
int smt_sharded_ivfpq_load(smt_sharded_corpus *corpus, const char *path, smt_sharded_ivfpq **out);
int smt_sharded_ivfpq_append(smt_sharded_ivfpq *index, uint64_t *n_added);
int smt_sharded_ivfpq_info(const smt_sharded_ivfpq *index, uint64_t *rows_covered, uint32_t *nlist, uint64_t *index_bytes);

/* Exactness bookkeeping.  The f32 scan nominates top_k + 8 rows per list and the select stage PROVES per query
 * that no other row can belong to the exact answer (the k-th exact distance lies more than the f32 error bound
 * below the worst nominated f32 distance).  When the proof fails -- more than 8 near-ties around the k-th place --
 * smt_search / smt_sharded_search re-answer that query exhaustively; the *_device entry points cannot (nothing
 * synchronises): they count such queries here and, in their _ex form, say WHICH query of the call it was
 * (SMT_STATUS_*).  A batched query whose candidate buffer overflowed is reported the same way (the device entry points
 * deliver its list as it is; the host entry points re-answer it).  reset != 0 clears the counter.
 * (smt_ivfpq_search_device has no _ex form: an IVF answer carries no certificate -- exact distances, approximate
 * membership by contract -- so there is no per-query verdict to report.) */
int smt_ctx_uncertain_count(smt_ctx *ctx, uint64_t *count, int reset);

/* Test hook for the kept range sets: a range list (the path subset of a workspace search) that a corpus sees for the second time
 * is kept on the device with its tile / chunk tables -- up to 4 lists per corpus, least recently used first out -- so that a session
 * searching the same file set again uploads only its queries.  kept: sets alive; hits: searches answered from one; builds: sets made. */
int smt_debug_range_sets(const smt_corpus *corpus, uint64_t *kept, uint64_t *hits, uint64_t *builds);

/* Test hook for delivered answers (tuning key direct_delivery): how many host-form searches on this context got their answer written
 * into pinned host memory by the select kernel and waited on its completion word, instead of a D2H copy + hipStreamSynchronize. */
int smt_debug_deliveries(smt_ctx *ctx, uint64_t *count);

/* Test hook for the SPMD error paths of multi-process groups: arms ONE injected failure with status `code` (an SMT_E_* value) on
 * THIS process's ranks; it fires at the next step of kind `where` and disarms.  The tests arm it on one rank of an n-rank group and
 * check that every rank returns `code` from the same call and that none is left waiting inside a collective.
 *   SMT_DEBUG_FAIL_STAGE  the local scan + select stage of the next smt_sharded_search that exchanges packed k-lists
 *   SMT_DEBUG_FAIL_AGREE  the next status agreement (threshold / large-k searches, sharded save / append / embed / index life cycle)
 *   SMT_DEBUG_FAIL_BUILD  the set-up of the next shared-centroid smt_sharded_ivfpq_build (before its first all-reduce)
 * where == 0 disarms. */
#define SMT_DEBUG_FAIL_STAGE 1
#define SMT_DEBUG_FAIL_AGREE 2
#define SMT_DEBUG_FAIL_BUILD 3
int smt_debug_group_fail_next(smt_group *group, int where, int code);

/* Test hook for the certificate's error bound: the f32 distances the batched kernels NOMINATE candidates with
 * (f32 MFMA, or bf16 x 3 split products when tuning key gemm_bf16x3 is set -- the default), for nq <= 32 host
 * queries against rows [first_row, first_row + n_rows) of the corpus; out is a host buffer [n_rows][32].
 * These values never reach an answer: results carry exact (f64) distances. */
int smt_debug_batched_scores(smt_corpus *corpus, const float *queries, uint32_t nq, uint64_t first_row, uint32_t n_rows,
                             float *out);

/* The context's second stream (hipStream_t), created on first use: async selects run on it.  A host that
 * chains more work behind an async select (an RCCL all-gather of its output, the merge of the gathered
 * lists with tuning key merge_on_aux) enqueues it here so that the main stream carries nothing but scans. */
int smt_ctx_aux_stream(smt_ctx *ctx, void **stream_out);

/* Tuning knobs; for benchmarking sweeps and throughput pipelines.  Keys:
 *   scan_blocks, scan_threads, scan_unroll (2/4/8), scan_nontemporal   K2 launch shape
 *   scan_steal (0/1/2/4/8/16), scan_steal_pct (1..50)   K2, unfiltered: the last scan_steal_pct per cent of the rows are dealt to the
 *                        blocks WHILE the kernel runs, in groups of scan_steal rounds (0, the default: the static deal).  Evens the
 *                        blocks' finish times out and gains nothing: the launch is bound by the aggregate read rate (A/B only)
 *   gemm_blocks, gemm_qsplit, gemm_ldsrow, gemm_dma_nt                     K3 (f32 kernels / range-filtered batches)
 *   gemm_bootstrap (1)                                                     K3 row-register kernel: first thresholds from a bootstrap level of
 *                                                                          tile minima (0: the appended-levels plan of rounds 1-3; A/B only)
 *   fallback_batch_min_rows   two or more uncertain queries of one call on a shard of at least this many rows (100 000)
 *                        are re-answered by ONE batched threshold pass instead of one exhaustive scan each
 *   guard_band (8..56)   the scans nominate min(64, top_k + guard_band) rows per list; the exactness proof tolerates that
 *                        many near-ties around the k-th place before a query has to be re-answered exhaustively
 *   gemm_min_nq, gemm_min_rows_small   from this many queries (default 3; 8+ always) on shards of this many rows (1 M)
 *                        a call takes the batched path K3 instead of K2 passes of <= 4 queries (2 queries: 4 x the rows)
 *   gemm_rowreg (1/0)    unfiltered batches run gemm_rowreg_kernel (coalesced row loads + LDS transpose)
 *   gemm_nominate (0/1/2/3) how the row-register kernel nominates: 1 bf16 x 3 (band 7e-5), 2 f16 x 2 (a third fewer MFMAs,
 *                        band 5.2e-4, guard band >= 16), 3 f16 x 1 (a third of the MFMAs, band 1.0e-3, guard band >= 24);
 *                        0 = on shards <= 32 M rows f16 x 1 from 129 queries, f16 x 2 at 128, bf16 x 3 below; with the corpus'
 *                        operand image: f16 x 1 from 128 queries, f16 x 2 below, shards up to 2^28 rows
 *   corpus_image (1/0)   corpora the library owns build and keep their fp16 operand image (smt_corpus_prepack)
 *   gemm_image (1/0)     batched searches read the image when the corpus has one
 *   image_scan_min_rows  shards of at least this many rows (1 500 000; 0 = never) answer ONE query from their image as well
 *                        (3..7 queries from two thirds of that), and build it at their fourth small search if they have none
 *   image_use_min_rows   a shard that already HAS its image answers one query from it from this many rows (400 000), two from
 *                        1/5 of it, three and more from 1/60 (unfiltered calls; 0 = only the rule above)
 *   gemm_boot_fine (1/0) corpora of 2 Ki .. 32 Ki tiles take their first thresholds from every 2nd / 4th / 8th tile (0: every 16th)
 *   gemm_buffered (1/0)  A/B switch of the LDS nomination buffer
 *   gemm_split_last (0/1/2)  levels run in two parts with a select pass in between: 0 none, 1 a ratio-16 last level, 2 (default)
 *                        also the first level after the bootstrap
 *   embed_batched (bit mask, default 3)  K1: bit 0 batched id loads, bit 1 the id-prefetch kernel, bit 2 a test hook (64-token
 *                        spans), bit 3 runs of equal line counts instead of equal work (A/B); 0 = the round-2 kernel.  NOTE: 1 no
 *                        longer means "everything on" -- pass 3
 *   gemm_resident        accepted and ignored (its kernel left in round 2)
 *   gemm_bf16x3 (1/0)    K3 nominates with bf16 x 3 split products on the bf16 MFMA pipe (default) or with f32 MFMAs;
 *                        answers are identical either way (exact re-scoring + the exactness certificate)
 *   direct_delivery (1/0)  smt_search top-k calls with a small answer (<= 32 queries, <= 8 KiB of rows and distances) get it DELIVERED
 *                        by the select kernel: its last block writes the answer into pinned host memory and a completion word behind
 *                        it, which the host waits on -- no D2H copy command, no hipStreamSynchronize (~10 us of a small call); 0: A/B
 *   prof_select (0/1), prof_every (N: HIP events on one launch in N)       profiling cost control
 *   scan_debug_ptr, select_debug_ptr                                       device pointers for phase stamps
 *   async_select (0/1)   smt_search_topk_device with ONE query: the select stage of call i runs on an
 *                        internal second stream WHILE the scan of call i+1 runs (device-scope flags between
 *                        the two kernels).  Outputs are complete after smt_ctx_synchronize (or any other
 *                        call on the context, which drains the pipeline first).  Throughput mode for
 *                        back-to-back single queries; off by default.
 *   merge_on_aux (0/1)   smt_merge_topk_packed_device is enqueued on the aux stream (see smt_ctx_aux_stream) */
int smt_set_tuning(smt_ctx *ctx, const char *key, int64_t value);

/* ------------------------------------------------------------------ host ids
 * FNV-1a 64 and the point ids of the workspace store
 * (src/workspace/store.rs:651-661, :82-89, :75-80) -- pure host helpers. */
uint64_t smt_fnv1a_hash(const uint8_t *bytes, uint64_t n);
uint64_t smt_line_embedding_id(const char *path, int32_t line_number);
uint64_t smt_doc_meta_id(const char *path);

#ifdef __cplusplus
}
#endif
#endif /* SEMTOOLS_HIP_H */
