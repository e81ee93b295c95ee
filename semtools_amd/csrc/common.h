// common.h -- internal declarations shared by the libsemtools_hip.so sources.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/semtools_hip.h"

namespace smt {

void set_error(const char *fmt, ...);

#define SMT_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            smt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                           __FILE__, __LINE__);                                          \
            return SMT_E_HIP;                                                            \
        }                                                                                \
    } while (0)

#define SMT_REQUIRE(cond, msg)                                                           \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            smt::set_error("invalid argument: %s (%s)", msg, #cond);                     \
            return SMT_E_INVALID;                                                        \
        }                                                                                \
    } while (0)

// bits of smt_ctx::attr_done
enum : uint32_t { ATTR_SELECT = 1u, ATTR_GEMM = 2u, ATTR_IVF_ASSIGN = 4u, ATTR_IVF_SCORE = 8u, ATTR_GEMM_LDS = 16u, ATTR_IVF_LPCA = 32u };

// One candidate on the device: key = (f32 distance bits << 32) | local row.
// Distances are clipped to >= 0, so the IEEE bit pattern is order-preserving
// and ascending u64 order == (distance asc, row asc).
typedef unsigned long long key_t64;
static constexpr key_t64 KEY_PAD = 0xFFFFFFFFFFFFFFFFull;

struct ProfEntry {
    std::vector<hipEvent_t> ev;  // start/stop pairs, grown on demand
    size_t used = 0;
    uint64_t calls = 0;          // prof_begin calls since the last reset (sampling: tuning key prof_every)
    bool armed = false;          // the current begin/end pair is being recorded
};

struct Tuning {
    int scan_blocks = 0;        // 0 = CU count (one block list per CU feeds the select stage)
    int scan_threads = 512;     // 8 waves/CU x (4 + 4 prefetched) rows = 64 KiB in flight per CU (swept on MI355X)
    int scan_unroll = 4;
    int scan_nontemporal = 1;
    int scan_prefetch = 1;      // software-pipelined row loads (single-query kernel)
    int scan_steal = 0;         // K2, unfiltered: rounds per dynamically dealt group (power of two; 0 = the static deal, the default: the dynamic deal evens the blocks out and gains nothing -- the launch is bound by the part's aggregate read rate): scan_kernels.hip STEAL
    int scan_steal_pct = 6;     // ... and the share of the corpus dealt that way, at the end of the launch (1..50 per cent)
    int gemm_blocks = 0;        // 0 = CU count
    int prof_every = 1;         // HIP events bracket one launch in N (an event pair costs ~6 us of stream time)
    int merge_on_aux = 0;       // 1: smt_merge_topk_packed_device runs on the aux stream (behind the async select it consumes)
    int async_select = 0;       // 1: single-query top-k searches overlap their select stage with the next scan
    int gemm_image = 1;         // 1: batched searches read the corpus' fp16 operand image when it has one (0: A/B only)
    int64_t image_scan_min_rows = 1500000;   // shards this large answer ONE query from the operand image too, when they have one; 2..7 queries from a third of this (0: never)
    int64_t image_use_min_rows = 400000;     // a corpus that already HAS its image answers one query from it from this many rows (unfiltered calls), two from 1/5 of it, three and more from 1/60 (0: only the image_scan_min_rows rule)
    int corpus_image = 1;       // 1: a corpus this library owns builds its operand image at the first batch of >= 8 queries (>= 64 Ki rows)
    int embed_batched = 3;      // bit 0: K1 finishes parked lines wave-wide; bit 1: token ids prefetched one step ahead (A/B only)
    int gemm_split_last = 2;    // a streamed K3 sweep runs a thin-threshold level in two parts with a select pass in between: 1 = its last level (ratio >= 16), 2 = also the quarter-corpus level of a bootstrap plan (0: none; A/B)
    int gemm_boot_fine = 1;     // 1: corpora of 2 Ki .. 32 Ki tiles bootstrap over every 2nd / 4th / 8th tile (0: every 16th, as up to 128 Ki tiles; A/B)
    int gemm_bootstrap = 1;     // 1: gemm_rowreg_kernel batches start from a bootstrap level of tile minima (0: the round-1..3 plan of appended levels; A/B only)
    int gemm_buffered = 1;      // 1: gemm_rowreg_kernel nominations go through the wave's LDS buffer (0: straight to the lists; A/B only)
    int gemm_qsplit = 1;        // 1: K3 levels with fewer row-tile groups than CUs split the query tiles over blocks
    int gemm_dma_nt = 1;        // 1: the LDS-row kernel's row DMA carries the nt cache policy (streamed once: 1.99 -> 1.84 ms at 32 x 10 M)
    int64_t fallback_batch_min_rows = 100000;   // >= 2 uncertain queries of a call on a shard this large are re-answered by ONE batched threshold pass
    int guard_band = 8;         // K2 / K3 nominate min(64, top_k + guard_band) rows per list (8..56)
    int gemm_min_nq = 5;        // batches of this many queries (up to 7) take K3 (document subsets: when rows x queries >= 1.2 x gemm_min_rows_small); 8+ always do; up to two queries fewer take it between 1/50 and 4/5 of gemm_min_rows_small rows
    int64_t gemm_min_rows_small = 1000000;   // K3 from gemm_min_nq queries when rows x queries >= 1.2 x this (search.cpp topk_dispatch)
    int gemm_nominate = 0;      // gemm_rowreg_kernel: 0 auto (shards <= 32 M rows: f16 x 2 from 128 queries, f16 x 1 from 256), 1 bf16 x 3, 2 f16 x 2, 3 f16 x 1
    int gemm_rowreg = 1;        // 1: with gemm_bf16x3, unfiltered batches use gemm_rowreg_kernel (coalesced row loads + LDS transpose)
    int gemm_bf16x3 = 1;        // 1: K3 nominates candidates with bf16 x 3 split products on the bf16 MFMA pipe (mfma_tile.h); 0: f32 MFMA
    int gemm_ldsrow = 1;        // 1: batches <= 128 queries and range-filtered batches use the LDS-row kernel (64 queries per pass)
    int prof_select = 1;        // 0: do not bracket the select stage with events (2 fewer event records per query)
    int direct_delivery = 1;    // 1: host-form top-k searches with small answers (<= 8 KiB) get them DELIVERED by the select kernel into pinned host memory, completion word last; the host waits on that word (search.cpp; 0: D2H copy + hipStreamSynchronize, A/B)
    int64_t scan_debug_ptr = 0;    // device pointer to (2*waves + blocks) u64 wall_clock64 stamps (profiling only)
    int64_t select_debug_ptr = 0;  // device pointer to 16 u64 for phase stamps (profiling only)
};

}  // namespace smt

struct smt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 0;
    // scratch (grown on demand, reused across calls)
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    uint64_t *d_embed_runs = nullptr;   // K1: first line of every group's run (token-balanced runs, embed_kernels.hip), made on first use
    size_t embed_runs_cap = 0;          // entries
    void *h_pinned = nullptr;
    size_t pinned_bytes = 0;
    void *h_pinned_in = nullptr;   // smt_search's inputs (queries, ranges, prefixes) assembled for ONE upload
    size_t pinned_in_bytes = 0;
    void *d_stage = nullptr;   // smt_search's per-call inputs/outputs (queries, ranges, result lists)
    size_t stage_bytes = 0;
    // async select (tuning key async_select): the select of query i runs on aux_stream WHILE query i+1 scans;
    // the two kernels meet through device-scope flags, not stream events
    hipStream_t aux_stream = nullptr;
    unsigned long long *d_status = nullptr; // [0] selects whose exactness certificate failed since the last read (sticky); [2], [3] the row domain check's (count, first row) (domain.hip); [4] the select blocks of a delivering launch that have finished (scan_kernels.hip)
    unsigned long long *d_flags = nullptr;  // [0] scan_done step, [1] select_done step, [2] blocks finished, [3] timeout flag
    uint64_t async_step = 0;
    bool async_pending = false;
    unsigned int *d_steal = nullptr;         // K2 STEAL: 64 group counters, one per launch in rotation (scan_kernels.hip)
    uint64_t steal_seq = 0;
    unsigned long long deliver_seq = 0;      // sequence number of the last delivered answer (its completion word in pinned memory)
    uint64_t deliveries = 0;                 // host-form searches answered that way (smt_debug_deliveries)
    bool prof_on = false;
    // kernels whose >64 KiB dynamic-LDS attribute has been set ON THIS DEVICE (bit per kernel family, smt::ATTR_*).
    // Per context, not per process: hipFuncSetAttribute applies to the current device's copy of the function, and a
    // group (group.cpp) owns one context per GPU inside one process.
    uint32_t attr_done = 0;
    std::map<std::string, smt::ProfEntry> prof;
    smt::Tuning tune;
};

namespace smt {
// A range list (the path subset of a workspace search, src/workspace/store.rs:495-515) staged on the device and KEPT on the corpus:
// a `serve -w` session, an `ask` agent or a shell loop search the same file set again and again, and validating + prefixing +
// uploading 5000 ranges and rebuilding their tile table was 0.17 ms of a 0.59 ms one-query call.  Identified by two independent
// 64-bit hashes of the list; built the SECOND time a list is seen (one-shot subsets never allocate); at most RANGE_SETS_MAX per
// corpus, least recently used goes first.  The hashes only FIND a candidate: the kept host copy of the list decides (memcmp).  The tables depend on the ranges only, never on the rows: appends do not invalidate them.
struct RangeSet {
    uint64_t h1 = 0, h2 = 0;
    uint32_t n_in = 0;                 // ranges as passed (empty ones included: part of the identity)
    uint32_t nr = 0;                   // non-empty ranges on the device
    uint64_t n_virtual = 0, n_chunks = 0, n_vtiles = 0, max_end = 0;
    char *dev = nullptr;               // [ranges | prefix | chunk_prefix | tile_prefix | tile table | chunk table]
    smt_range *d_r = nullptr;
    uint64_t *d_p = nullptr, *d_cp = nullptr, *d_tp = nullptr;
    uint64_t *d_tile_table = nullptr, *d_chunk_table = nullptr;   // nullptr: too large to keep (rebuilt per call in scratch)
    bool have_tile_table = false, have_chunk_table = false;
    uint64_t last_use = 0;
    std::vector<smt_range> host_ranges;   // the list as passed (n_in entries): a hash hit is confirmed by comparing it
};
struct FileWriter;
constexpr int RANGE_SETS_MAX = 4;
constexpr size_t RANGE_SET_TABLE_BYTES_MAX = (size_t)96 << 20;   // tile + chunk table of one set
}  // namespace smt

struct smt_corpus {
    smt_ctx *ctx = nullptr;
    float *d_rows = nullptr;
    uint64_t rows = 0;
    uint64_t capacity = 0;
    uint32_t dim = 0;
    bool owned = true;
    // fp16 OPERAND IMAGE (gemm_rowreg.hip: pack_image_kernel): what the batched fp16 nomination modes multiply with, kept
    // beside the f32 rows -- 512 B per row, 16 KiB per 32-row tile -- so that a batch reads half the bytes and converts nothing.
    // Derived data: the f32 rows stay the truth (every returned distance is re-scored from them).  image_rows = the rows the
    // image describes; appends grow `rows` past it, smt_corpus_write_rows pulls it back: corpus_image_sync packs what is missing.
    void *image = nullptr;
    uint32_t *image_zero = nullptr;
    uint64_t image_cap_tiles = 0;
    uint64_t image_rows = 0;
    uint32_t small_searches = 0; // searches of < 8 queries seen while the shard was large enough to scan its image (search.cpp topk_dispatch)
    int image_mode = 0;          // 0: by policy (tuning key corpus_image; owned corpora only), 1: requested (smt_corpus_prepack), -1: refused
    struct smt::FileWriter *writer = nullptr;  // corpus_io.cpp: the background writer of SMT_APPEND_WRITE_AHEAD
    std::vector<smt::RangeSet *> range_sets;   // search.cpp: ranges_on_device
    uint64_t range_clock = 0;
    uint64_t range_seen[16][2] = {};           // hashes of lists seen once (a set is built on the second sight)
    uint32_t range_seen_next = 0;
    uint64_t range_set_hits = 0, range_set_builds = 0;
};

struct smt_model {
    smt_ctx *ctx = nullptr;
    float *d_table = nullptr;
    uint64_t V = 0;
    uint32_t D = 0;
    int normalize = 1;
    bool owned = true;
};

namespace smt {

int check_ctx(const smt_ctx *ctx);
int bind_device(smt_ctx *ctx, bool drain = true);   // hipSetDevice + (drain) wait for async selects
// The handler of every int-returning extern "C" entry point (each is a function-try-block): a C++ exception -- bad_alloc from a host
// vector, anything a callee throws -- becomes an error code and message; it must never unwind into a C, Rust or ctypes caller.
int api_catch() noexcept;
int corpus_reserve(smt_corpus *c, uint64_t rows_needed);
// the image up to date for a batch of nq queries, or left alone (policy, memory): sets *image / *image_zero (nullptr = none)
int corpus_image_sync(smt_corpus *c, uint32_t nq, const void **image, const uint32_t **image_zero);
void corpus_image_drop(smt_corpus *c);
int launch_pack_image(smt_ctx *ctx, const float *corpus, uint64_t n_rows, uint64_t first_tile, uint64_t n_tiles, void *image,
                      uint32_t *image_zero);
int ensure_scratch(smt_ctx *ctx, size_t bytes);
int ensure_pinned(smt_ctx *ctx, size_t bytes);
int ensure_pinned_in(smt_ctx *ctx, size_t bytes);
int ensure_stage(smt_ctx *ctx, size_t bytes);
// Wait for select kernels still running on the aux stream (no-op unless async_select was used).  Every entry
// point that touches the context's scratch or reads results on the main stream calls this first.
int drain_async(smt_ctx *ctx);
int ensure_async(smt_ctx *ctx);  // aux stream + flags

// RAII-less helpers for event timing around a kernel family.
void prof_begin(smt_ctx *ctx, const char *name);
void prof_end(smt_ctx *ctx, const char *name);
// ... on a given stream of the context's device (the pair may straddle two streams: "exchange" begins behind a rank's own select and
// ends in front of the merge that waited for every other rank's list)
void prof_begin_on(smt_ctx *ctx, const char *name, hipStream_t st);
void prof_end_on(smt_ctx *ctx, const char *name, hipStream_t st);

// ---- kernel launchers (defined in the .hip files) -------------------------
// K2: single/few-query f32 scan with per-wave top-k' lists, then merge +
// f64 rescoring.  All pointers device.  Results: out_rows[nq][k_out],
// out_dist[nq][k_out], out_counts[nq] (device).
struct ScanArgs {
    const float *corpus;      // [rows x 256]
    uint64_t rows;            // rows in this shard (< 2^32)
    const float *queries;     // device [nq x 256]
    uint32_t nq;
    uint32_t k_out;           // results wanted per query
    const smt_range *ranges;  // device, or nullptr
    const uint64_t *range_prefix;  // device exclusive prefix of range lengths [n_ranges+1] (large-k path)
    const uint64_t *range_chunk_prefix;  // device exclusive prefix of ceil(len / FILTER_CHUNK) [n_ranges+1]
    uint32_t n_ranges;
    uint64_t n_virtual;       // rows to scan (sum of ranges, or rows)
    uint64_t n_chunks;        // with ranges: total chunks = range_chunk_prefix[n_ranges]
    const uint64_t *range_tile_prefix = nullptr;  // device exclusive prefix of the aligned 32-row tiles each range is the FIRST to touch [n_ranges+1]
    uint64_t n_vtiles = 0;    // with ranges: tiles holding at least one wanted row = range_tile_prefix[n_ranges]
    int ws_threshold;         // 1: apply score > thr_score (f32) in the final stage
    float ws_thr_score;
    uint64_t row_base;
    uint64_t *out_rows;
    double *out_dist;
    uint64_t *out_counts;     // may be nullptr
    uint64_t *out_uncertain = nullptr;  // device [nq] or nullptr: non-zero where the f32 nomination could not be PROVEN to
                                        // contain the exact top-k (see SelectArgs::f32_err; SMT_STATUS_* code); the host API
                                        // then re-answers that query exhaustively
    uint32_t *out_status = nullptr;     // device [nq] or nullptr: the same verdict as a SMT_STATUS_* code per query -- what the
                                        // *_device_ex entry points hand their caller (include/semtools_hip.h)
    bool allow_async = false; // the caller does not read the outputs on the main stream before smt_ctx_synchronize
    const void *image = nullptr;          // the corpus' fp16 operand image covering all `rows` (smt_corpus::image), or nullptr
    const uint32_t *image_zero = nullptr;
    uint64_t out_stride = 0;  // words between the output lists of consecutive queries (0 = k_out); the packed
                              // [nq][2][k] exchange layout of group.cpp uses 2*k with out_dist = out_rows + k
    struct RangeSet *range_set = nullptr;   // the ranges come from a kept set: its tile / chunk tables are built once and reused
    const struct Delivery *deliver = nullptr;   // the select stage delivers the answers to pinned host memory (below)
};
int launch_scan_topk(smt_ctx *ctx, const ScanArgs &a);
void corpus_range_sets_drop(smt_corpus *c);   // search.cpp (smt_corpus_destroy)
// the kept set's table when there is one (built on first use), else `scratch_table` built for this call
int range_tile_table(smt_ctx *ctx, const ScanArgs &a, uint64_t *scratch_table, const uint64_t **table);
int range_chunk_table(smt_ctx *ctx, const ScanArgs &a, uint64_t *scratch_table, const uint64_t **table);

// Range-filtered scans stream FILTER_CHUNK-row chunks described by a per-chunk table (scan_kernels.hip).
constexpr int FILTER_CHUNK = 4;
int launch_build_chunk_table(smt_ctx *ctx, const smt_range *ranges, const uint64_t *chunk_prefix, uint32_t n_ranges,
                             uint64_t n_chunks, uint64_t *table);
// ... and, for gemm_rowreg_kernel, the aligned 32-row tiles they touch: tile | row mask << 32 per tile (scan_kernels.hip)
int launch_build_tile_table(smt_ctx *ctx, const smt_range *ranges, const uint64_t *tile_prefix, uint32_t n_ranges, uint64_t n_vtiles,
                            uint64_t *table);
// A filtered batch takes gemm_rowreg_kernel (and the operand image) when the wanted rows fill at least a quarter of the tiles they
// touch; sparser subsets (documents of a few lines, every tenth one wanted) keep the LDS-row kernel, which gathers 4-row chunks.
inline bool tiles_dense(uint64_t n_virtual, uint64_t n_vtiles) { return n_vtiles > 0 && n_vtiles * 32 <= 4 * n_virtual; }

// K4 (threshold.hip): every row with distance < max_distance, ordered (distance asc, row asc).
// Returns pointers into the context's pinned staging buffer, valid until the next call on the context.
struct ThresholdQuery {
    const float *corpus;
    uint64_t rows;
    const float *query;       // device [256]
    const smt_range *ranges;
    const uint64_t *range_chunk_prefix;  // see ScanArgs
    uint32_t n_ranges;
    uint64_t n_virtual;
    uint64_t n_chunks;
    double max_distance;
};
int run_threshold_query(smt_ctx *ctx, const ThresholdQuery &t, const uint32_t **rows_host, const double **dist_host,
                        uint64_t *n_pass);
int launch_rescore_rows(smt_ctx *ctx, const float *corpus, const float *query, const uint32_t *rows,
                        uint64_t n, double *out_dist);
int launch_gather_rows256(smt_ctx *ctx, const float *src, const uint32_t *idx_dev, uint32_t n, float *dst);   // dst[i] = src[idx[i]]
// ... of many queries at once: rows[i] against query qidx[i] of the [.][256] block (the batched exhaustive re-answer)
int launch_rescore_rows_multi(smt_ctx *ctx, const float *corpus, const float *queries, const uint32_t *rows, const uint32_t *qidx,
                              uint64_t n, double *out_dist);

// DELIVERY of a small answer by the select kernel itself (host-form searches, search.cpp): the outputs of a launch lie in ONE device
// block (dev_out, n_words 8-byte words); the LAST select block to finish copies it to pinned host memory (host_out) and then stores
// `seq` into the pinned word host_flag with a system-scope release.  The host waits on that word instead of enqueueing a D2H copy
// and calling hipStreamSynchronize: measured on an MI355X (tools/micro/call_floor.hip, profiles/r06_call_floor.json) a kernel that
// announces its result in pinned memory is back in 6.5 us, kernel + hipStreamSynchronize in 12.5, with the download 15+.
struct Delivery {
    const unsigned long long *dev_out = nullptr;
    unsigned long long *host_out = nullptr;
    uint32_t n_words = 0;
    unsigned long long *host_flag = nullptr;
    unsigned long long seq = 0;
    unsigned long long *done = nullptr;   // device counter of finished select blocks (0 between launches)
};

// Select stage: block lists -> best k_out per query with exact f64 distances (scan_kernels.hip).
struct SelectArgs {
    const float *corpus = nullptr;
    const float *queries = nullptr;
    uint32_t nq = 0;
    key_t64 *lists = nullptr;      // query qi's lists start at lists + qi * list_stride: [n_lists][kp]
    uint32_t n_lists = 0;
    uint32_t kp = 0;
    uint64_t list_stride = 0;
    uint32_t k_out = 0;
    int ws_threshold = 0;
    float ws_thr_score = 0.f;
    uint64_t row_base = 0;
    uint64_t *out_rows = nullptr;
    double *out_dist = nullptr;
    uint64_t *out_counts = nullptr;
    uint64_t async_step = 0;       // != 0: on the aux stream, gated by the flags
    uint64_t out_stride = 0;       // 0 = k_out
    // Exactness certificate.  The lists hold the kp best rows by the SCAN's f32 distance; every other row has
    // d32 >= tau32 (the kp-th nominated), hence exact distance >= tau32 - f32_err when f32_err bounds the scan
    // kernel's |d32 - d64|.  If the k_out-th exact distance is below that, no row outside the lists can belong to
    // the answer: it is the exact top-k.  Otherwise the query is flagged (out_uncertain[q] = 1, the context's
    // sticky counter is bumped) and the host entry points re-answer it exhaustively.  0 = no certificate (IVF-PQ).
    double f32_err = 0.0;
    uint64_t *out_uncertain = nullptr;
    uint32_t *out_status = nullptr;   // [nq] or nullptr: 0 proved exact, 1 certificate failed, 2 a candidate buffer overflowed
    // batched path: != 0 where a query's candidate buffer overflowed between two level selects (rows were dropped: the lists may
    // miss an answer row).  Such a query is flagged like a failed certificate -- the host entry points re-answer it exhaustively,
    // the device entry points count it -- so that no batched call needs a host synchronisation of its own (until round 5
    // launch_gemm_topk read this flag back after every batch).
    const unsigned int *overflow = nullptr;
    const Delivery *deliver = nullptr;   // the last block carries the answers home (not with async_step)
};
int launch_select(smt_ctx *ctx, const SelectArgs &s);


// |f32 scan distance - exact distance| bounds used for the certificate.  What an MFMA does to its accumulator was MEASURED
// on gfx950 (tools/micro/mfma_rounding.hip, profiles/r03_mfma_rounding.json): v_mfma_f32_32x32x2_f32 is a chain of fused
// multiply-adds, each rounded to nearest-even; the 16-bit 32x32x16 MFMAs form their 16 products exactly, align them with the
// accumulator keeping 5 bits below its ulp, and round the sum ONCE to nearest-even -- at most 2 ulp of the accumulator per
// instruction (measured on positive 256-dim dots: 2 ulp in total over 16 instructions).
//   K2/K4: 4 FMAs per lane + a reduction tree of 6 levels (wave_sum4: xor 1, xor 2, rotate 4, rotate 8, +-16, +-32; wave_sum: four
//   DPP levels and two scalar ones) -- the bound below was taken for an 8-step tree: <= 12 roundings of terms whose absolute sum is
//   <= 1, plus two rsqrt/multiplies: < 1e-6.
//   f32 MFMA: 256 products accumulated one by one, each add rounded to nearest: <= 256 * 2^-24 = 1.5e-5.
constexpr double F32_ERR_SCAN = 4e-6;
constexpr double F32_ERR_MFMA = 2e-5;
// bf16 x 3 split products (mfma_tile.h): representation 3 * 2^-16 |x||q| (4.6e-5: the dropped lo.lo term and the two
// residuals, each <= 2^-16 sum |x_i q_i|) + 48 instructions x 2 ulp (1.1e-5) + the f32 scaling by 1/|x|, 1/|q| (< 1e-6)
// = 5.8e-5.  Constructed worst case (tests/test_gpu_batched.py): 2.6e-5; random corpora: 1.0e-5.
constexpr double F32_ERR_BF16X3 = 7e-5;
// f16 x 2: ONE fp16 operand for the rows: 2^-11 sum |x_i q_i| <= 4.88e-4 (the query, hi + lo, carries 22 bits) + 32
// instructions x 2 ulp (7.6e-6) + scaling = 4.97e-4.  Constructed worst case: 3.7e-4; random corpora: 2.3e-4.
constexpr double F32_ERR_F16X2 = 5.2e-4;
// f16 x 1: the query is a single fp16 operand too: 2 x 2^-11 (+ 2^-22) = 9.77e-4 + 16 instructions x 2 ulp (3.8e-6) = 9.8e-4.
constexpr double F32_ERR_F16X1 = 1.0e-3;
// K2/K3 keep k + 8 <= 64 candidates per list: top_k above this goes to the all-keys path (largek.hip)
constexpr uint32_t SCAN_MAX_K = 56;

// per-query result list of the host-side search (search.cpp)
struct LocalHits {
    std::vector<uint64_t> rows;  // global rows (row_base added)
    std::vector<double> dist;
};
int search_local_host(smt_corpus *corpus, const float *queries, uint32_t nq, uint32_t top_k, double max_distance, int mode,
                      const smt_range *ranges, uint32_t n_ranges, uint64_t row_base, std::vector<LocalHits> &out);
// a zero query in workspace mode has a constant answer (search.cpp: qdrant scores it 0 against every point)
bool query_is_zero(const float *q);
void workspace_zero_query_hits(const smt_range *ranges, uint32_t n_ranges, uint64_t n_rows, uint32_t top_k, bool has_thr, double max_distance,
                               uint64_t row_base, LocalHits &out);
int deliver_hits(const std::vector<LocalHits> &hits, uint64_t *out_rows, double *out_dist, uint64_t *out_counts, uint64_t out_cap);
int search_topk_packed_local(smt_corpus *corpus, const float *queries_dev, uint32_t nq, uint32_t k_pad, int ws_threshold,
                             float ws_thr_score, const smt_range *ranges_local, uint32_t n_ranges, bool filtered,
                             uint64_t row_base, uint64_t *packed_dev, uint64_t *uncertain_dev, bool allow_async);
// corpus file slices (corpus_io.cpp): shared by smt_corpus_save/load and the sharded corpus of group.cpp
int corpus_file_info(const char *path, uint64_t *rows, uint32_t *dim);
int corpus_load_slice(smt_corpus *c, const char *path, uint64_t first_row, uint64_t n_rows);
int corpus_file_begin(const char *path, uint32_t dim, uint64_t total_rows);
int corpus_save_slice(smt_corpus *c, const char *path, uint64_t file_first_row);
struct FileRun { uint64_t local_first, n_rows, file_first_row; };
int corpus_save_runs(smt_corpus *c, const char *path, const FileRun *runs, size_t n_runs, bool durable = true);
int corpus_append_to_file_ex(smt_corpus *c, const char *path, uint64_t rows_on_disk, uint64_t rows_written, int flags);
// the background writer of write-ahead appends: wait for it before the rows it reads move, change or go (and before a commit)
int corpus_writer_drain(smt_corpus *c, std::string *failed = nullptr);
void corpus_writer_destroy(smt_corpus *c);
int corpus_file_extend(const char *path, uint32_t dim, uint64_t expect_rows, uint64_t new_rows);
int corpus_file_commit(const char *path, uint64_t rows);
int launch_merge_topk_packed_on(smt_ctx *ctx, hipStream_t st, const uint64_t *packed, uint32_t n_lists, uint32_t nq,
                                uint32_t k_in, uint32_t k_out, uint64_t *out_packed, uint64_t list_stride_words = 0);

// merge of lists read where they lie (one base pointer per list): the peer transport of group.cpp
#define SMT_MAX_MERGE_SOURCES 64
struct MergeSources { const uint64_t *list[SMT_MAX_MERGE_SOURCES]; };
int launch_merge_topk_sources_on(hipStream_t st, const MergeSources &src, uint32_t n_lists, uint32_t nq, uint32_t k_in, uint32_t k_out,
                                 uint64_t *out_packed);
// out[q] = max over the n sources of their q-th status word (u64 words, SMT_STATUS_* codes): the per-query verdict of a sharded
// search is the worst of its shards'.  Sources by pointer (peer transport), or `stride_words` apart behind `base` (gathered buffer).
int launch_combine_status_on(hipStream_t st, const MergeSources *src, const uint64_t *base, uint64_t stride_words, uint32_t n, uint32_t nq,
                             uint32_t *out);

int launch_merge_topk(smt_ctx *ctx, const uint64_t *rows, const double *dist, uint32_t n_lists,
                      uint32_t nq, uint32_t k_in, uint32_t k_out, uint64_t *out_rows,
                      double *out_dist);

int launch_merge_topk_packed(smt_ctx *ctx, const uint64_t *packed, uint32_t n_lists, uint32_t nq, uint32_t k_in,
                             uint32_t k_out, uint64_t *out_packed);

// top_k > 64 (rare): all keys + radix sort + exact rescoring of the best n_cand rows
int launch_largek_candidates(smt_ctx *ctx, const float *corpus, const float *query_dev, const smt_range *ranges_dev,
                             const uint64_t *prefix_dev, uint32_t n_ranges, uint64_t n_virtual, uint64_t n_cand,
                             std::vector<uint32_t> &rows_out, std::vector<double> &dist_out, float *next_d32 = nullptr
                             /* f32 distance of the best row NOT among the candidates (+inf if none) */);

// IVF index as one rank of a shared-centroid build / packed search (ivfpq_build.hip / ivfpq_search.hip <-> group.cpp)
struct IvfBuildShare {
    uint32_t rank = 0, n_ranks = 1;
    // sum `sums` (n_sums int64, 2^-32 fixed point) and `counts` (n_counts u32) over the ranks, in place, enqueued on
    // the context's stream (ncclAllReduce), or synchronously for the copy transport
    int (*allreduce)(void *user, long long *sums_dev, size_t n_sums, unsigned int *counts_dev, size_t n_counts) = nullptr;
    // every rank reports the status of its set-up (allocations) and gets back the first failure of ANY rank: called once,
    // before the first allreduce, so that no rank enters a collective the others will never reach
    int (*agree)(void *user, int rc) = nullptr;
    void *user = nullptr;
};
}  // namespace smt
struct smt_ivfpq;
struct smt_ivfpq_params;
namespace smt {
int ivfpq_build_shared(smt_corpus *corpus, const smt_ivfpq_params *prm, const IvfBuildShare *share, smt_ivfpq **out);
int ivfpq_search_packed(smt_ivfpq *ix, const float *queries_dev, uint32_t nq, uint32_t top_k, uint32_t nprobe, uint32_t rerank,
                        uint64_t row_base, uint64_t *packed_dev);

// domain.hip: the numeric domain of rows and queries (finite, largest magnitude 0 or within [2^-40, 2^40]) and its checks
constexpr uint32_t DOMAIN_MIN_BITS = 0x2B800000u;   // 2^-40f
constexpr uint32_t DOMAIN_MAX_BITS = 0x53800000u;   // 2^40f
__host__ __device__ inline bool magnitude_in_domain(uint32_t max_abs_bits)
{
    return max_abs_bits == 0 || (max_abs_bits >= DOMAIN_MIN_BITS && max_abs_bits <= DOMAIN_MAX_BITS);
}
int check_rows_domain(smt_ctx *ctx, const float *d_rows, uint64_t n_rows, uint64_t *n_bad, uint64_t *first_bad);
int require_rows_domain(smt_ctx *ctx, const float *d_rows, uint64_t n_rows, const char *what, uint64_t base = 0);
int require_unit_rows(smt_ctx *ctx, const float *d_rows, uint64_t n_rows, const char *what, uint64_t base = 0);   // (the IVF index)
int64_t first_outside_domain_host(const float *v, uint64_t n, uint32_t dim);
int require_queries_domain_host(const float *queries, uint32_t nq, const char *what);

// K1
int launch_embed(smt_ctx *ctx, const float *table, uint64_t V, int normalize, const uint32_t *ids,
                 const uint64_t *offsets, uint64_t n_lines, uint32_t max_tokens, float *out,
                 uint64_t n_tokens_known /* total tokens of the batch when the host has the offsets, else 0 */);

// K3: batched queries, f32 MFMA with fused candidate selection.
int launch_gemm_topk(smt_ctx *ctx, const ScanArgs &a);
// bf16 hi / lo split image (mfma_tile.h layout) of n rows of 256 f32, padded with zero rows to n_pad (gemm_topk.hip)
int launch_split_rows_bf16(smt_ctx *ctx, const float *rows, uint32_t n, uint32_t n_pad, uint32_t *out);
// batched threshold pass (gemm_topk.hip): rows with nominating distance <= tau[q], per query, in scratch buffers
int launch_gemm_threshold(smt_ctx *ctx, const float *corpus, uint64_t rows, const void *image, const uint32_t *image_zero,
                          const float *queries, uint32_t nq, const float *tau, const key_t64 **cand_out,
                          const unsigned int **counts_out, uint32_t *cand_stride);
// test hook: the nominating f32 distances of the K3 kernels for <= 32 queries (gemm_topk.hip)
int launch_gemm_debug_scores(smt_ctx *ctx, const float *corpus, uint64_t first_row, uint32_t n_rows, const float *queries,
                             uint32_t nq, float *out);

}  // namespace smt
