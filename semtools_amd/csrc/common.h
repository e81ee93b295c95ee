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
    int gemm_blocks = 0;        // 0 = CU count
    int prof_every = 1;         // HIP events bracket one launch in N (an event pair costs ~6 us of stream time)
    int merge_on_aux = 0;       // 1: smt_merge_topk_packed_device runs on the aux stream (behind the async select it consumes)
    int async_select = 0;       // 1: single-query top-k searches overlap their select stage with the next scan
    int gemm_qsplit = 1;        // 1: K3 levels with fewer row-tile groups than CUs split the query tiles over blocks
    int gemm_resident = 1;      // 1: batches <= 128 queries use the resident-query, double-buffered-row kernel
    int prof_select = 1;        // 0: do not bracket the select stage with events (2 fewer event records per query)
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
    void *h_pinned = nullptr;
    size_t pinned_bytes = 0;
    void *d_stage = nullptr;   // smt_search's per-call inputs/outputs (queries, ranges, result lists)
    size_t stage_bytes = 0;
    // async select (tuning key async_select): the select of query i runs on aux_stream WHILE query i+1 scans;
    // the two kernels meet through device-scope flags, not stream events
    hipStream_t aux_stream = nullptr;
    unsigned long long *d_flags = nullptr;  // [0] scan_done step, [1] select_done step, [2] blocks finished, [3] timeout flag
    uint64_t async_step = 0;
    bool async_pending = false;
    bool prof_on = false;
    std::map<std::string, smt::ProfEntry> prof;
    smt::Tuning tune;
};

struct smt_corpus {
    smt_ctx *ctx = nullptr;
    float *d_rows = nullptr;
    uint64_t rows = 0;
    uint64_t capacity = 0;
    uint32_t dim = 0;
    bool owned = true;
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

int ensure_scratch(smt_ctx *ctx, size_t bytes);
int ensure_pinned(smt_ctx *ctx, size_t bytes);
int ensure_stage(smt_ctx *ctx, size_t bytes);
// Wait for select kernels still running on the aux stream (no-op unless async_select was used).  Every entry
// point that touches the context's scratch or reads results on the main stream calls this first.
int drain_async(smt_ctx *ctx);
int ensure_async(smt_ctx *ctx);  // aux stream + flags

// RAII-less helpers for event timing around a kernel family.
void prof_begin(smt_ctx *ctx, const char *name);
void prof_end(smt_ctx *ctx, const char *name);

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
    int ws_threshold;         // 1: apply score > thr_score (f32) in the final stage
    float ws_thr_score;
    uint64_t row_base;
    uint64_t *out_rows;
    double *out_dist;
    uint64_t *out_counts;     // may be nullptr
    bool allow_async = false; // the caller does not read the outputs on the main stream before smt_ctx_synchronize
};
int launch_scan_topk(smt_ctx *ctx, const ScanArgs &a);

// Range-filtered scans stream FILTER_CHUNK-row chunks described by a per-chunk table (scan_kernels.hip).
constexpr int FILTER_CHUNK = 4;
int launch_build_chunk_table(smt_ctx *ctx, const smt_range *ranges, const uint64_t *chunk_prefix, uint32_t n_ranges,
                             uint64_t n_chunks, uint64_t *table);

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

int launch_select(smt_ctx *ctx, const float *corpus, const float *queries, uint32_t nq, key_t64 *lists,
                  uint32_t n_lists, uint32_t kp, uint64_t list_stride, uint32_t k_out, int ws_threshold,
                  float ws_thr_score, uint64_t row_base, uint64_t *out_rows, double *out_dist,
                  uint64_t *out_counts, uint64_t async_step = 0 /* != 0: on the aux stream, gated by the flags */);

int launch_merge_topk(smt_ctx *ctx, const uint64_t *rows, const double *dist, uint32_t n_lists,
                      uint32_t nq, uint32_t k_in, uint32_t k_out, uint64_t *out_rows,
                      double *out_dist);

int launch_merge_topk_packed(smt_ctx *ctx, const uint64_t *packed, uint32_t n_lists, uint32_t nq, uint32_t k_in,
                             uint32_t k_out, uint64_t *out_packed);

// top_k > 64 (rare): all keys + radix sort + exact rescoring of the best n_cand rows
int launch_largek_candidates(smt_ctx *ctx, const float *corpus, const float *query_dev, const smt_range *ranges_dev,
                             const uint64_t *prefix_dev, uint32_t n_ranges, uint64_t n_virtual, uint64_t n_cand,
                             std::vector<uint32_t> &rows_out, std::vector<double> &dist_out);

// K1
int launch_embed(smt_ctx *ctx, const float *table, uint64_t V, int normalize, const uint32_t *ids,
                 const uint64_t *offsets, uint64_t n_lines, uint32_t max_tokens, float *out);

// K3: batched queries, f32 MFMA with fused candidate selection.
int launch_gemm_topk(smt_ctx *ctx, const ScanArgs &a);

}  // namespace smt
