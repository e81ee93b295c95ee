// search.cpp -- the search entry points of libsemtools_hip.so (include/semtools_hip.h: smt_search, smt_search_topk_device, the merge
// calls) and the host logic behind them: which kernel answers a call (topk_dispatch), the exactness fall-backs, threshold mode,
// delivery.  No CPU fallback: every path ends in a HIP kernel.
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <limits>
#include <string>
#include <vector>

#include <atomic>
#include <chrono>

#include "common.h"

using namespace smt;

namespace smt {

// The three exclusive prefixes a range-filtered search stages beside its ranges: rows before range i (large-k path), FILTER_CHUNK-row
// chunks before it (K2 / K4 / the LDS-row kernel) and aligned 32-row tiles before it (gemm_rowreg_kernel: a tile counts for the
// FIRST range that touches it).  host = [prefix | chunk_prefix | tile_prefix], each nr + 1 words.
static void range_prefixes(const std::vector<smt_range> &rr, std::vector<uint64_t> &host)
{
    const size_t nr = rr.size();
    host.assign(3 * (nr + 1), 0);
    uint64_t *prefix = host.data(), *chunk_prefix = prefix + nr + 1, *tile_prefix = chunk_prefix + nr + 1;
    uint64_t last_tile = UINT64_MAX;
    for (size_t i = 0; i < nr; ++i) {
        const uint64_t len = rr[i].end - rr[i].begin;
        prefix[i + 1] = prefix[i] + len;
        chunk_prefix[i + 1] = chunk_prefix[i] + (len + FILTER_CHUNK - 1) / FILTER_CHUNK;
        const uint64_t ft = rr[i].begin >> 5, lt = (rr[i].end - 1) >> 5;
        tile_prefix[i + 1] = tile_prefix[i] + (lt - ft + 1) - (ft == last_tile ? 1 : 0);
        last_tile = lt;
    }
}

// ---------------------------------------------------------------- kept range sets (common.h RangeSet)
static inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32;
    return x;
}

static void range_set_free(RangeSet *rs)
{
    if (!rs) return;
    if (rs->dev) (void)hipFree(rs->dev);
    delete rs;
}

void corpus_range_sets_drop(smt_corpus *c)
{
    for (RangeSet *rs : c->range_sets) range_set_free(rs);
    c->range_sets.clear();
}

// Ranges validated (as validate_ranges) and identified in one pass; the set kept for this list if there is one, else nullptr
// (with *build = true when the list has been seen before and deserves one now).
static int range_set_find(smt_corpus *corpus, const smt_range *ranges, uint32_t n, uint64_t *total, RangeSet **found, bool *build,
                          uint64_t *h1_out, uint64_t *h2_out)
{
    uint64_t prev_end = 0, t = 0, h1 = 0x9E3779B97F4A7C15ull ^ n, h2 = 0xC2B2AE3D27D4EB4Full + n;
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t b = ranges[i].begin, e = ranges[i].end;
        SMT_REQUIRE(b <= e, "range begin > end");
        SMT_REQUIRE(e <= corpus->rows, "range extends past the corpus");
        SMT_REQUIRE(i == 0 || b >= prev_end, "ranges must be sorted and disjoint");
        prev_end = e;
        t += e - b;
        h1 = (h1 ^ b) * 0x100000001B3ull; h1 = (h1 ^ e) * 0x100000001B3ull; h1 ^= h1 >> 29;
        h2 = (h2 + b) * 0x9FB21C651E98DF25ull; h2 = ((h2 << 31) | (h2 >> 33)) + e;
    }
    h1 = mix64(h1); h2 = mix64(h2 ^ t);
    *total = t;
    *h1_out = h1; *h2_out = h2;
    *found = nullptr;
    *build = false;
    for (RangeSet *rs : corpus->range_sets)
        if (rs->h1 == h1 && rs->h2 == h2 && rs->n_in == n && rs->n_virtual == t && rs->host_ranges.size() == n &&
            (n == 0 || memcmp(rs->host_ranges.data(), ranges, (size_t)n * sizeof(smt_range)) == 0)) {   // (a collision must not answer for another subset)
            rs->last_use = ++corpus->range_clock;
            ++corpus->range_set_hits;
            *found = rs;
            return SMT_OK;
        }
    for (auto &seen : corpus->range_seen)
        if (seen[0] == h1 && seen[1] == h2) { *build = true; return SMT_OK; }
    corpus->range_seen[corpus->range_seen_next % 16][0] = h1;
    corpus->range_seen[corpus->range_seen_next % 16][1] = h2;
    ++corpus->range_seen_next;
    return SMT_OK;
}

// A new kept set for (rr, prefixes): one device block, uploaded on the context's stream from pinned memory.
static int range_set_build(smt_corpus *corpus, const smt_range *ranges_in, uint32_t n_in, uint64_t h1, uint64_t h2,
                           const std::vector<smt_range> &rr, const std::vector<uint64_t> &prefixes, uint64_t n_virtual, RangeSet **out)
{
    smt_ctx *ctx = corpus->ctx;
    *out = nullptr;
    const uint32_t nr = (uint32_t)rr.size();
    const uint64_t n_chunks = prefixes[2 * (nr + 1) - 1], n_vtiles = prefixes[3 * (nr + 1) - 1];
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t r_bytes = al((size_t)nr * sizeof(smt_range)), p_bytes = (size_t)(nr + 1) * sizeof(uint64_t);
    const size_t head = r_bytes + al(3 * p_bytes);
    const bool keep_tables = (n_vtiles + n_chunks) * 8 <= RANGE_SET_TABLE_BYTES_MAX;
    const size_t b_tile = keep_tables ? al((size_t)n_vtiles * 8) : 0, b_chunk = keep_tables ? al((size_t)n_chunks * 8) : 0;
    if (corpus->range_sets.size() >= (size_t)RANGE_SETS_MAX) {
        // the least recently used set goes; kernels of earlier calls may still read it
        size_t lru = 0;
        for (size_t i = 1; i < corpus->range_sets.size(); ++i)
            if (corpus->range_sets[i]->last_use < corpus->range_sets[lru]->last_use) lru = i;
        SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->aux_stream) SMT_HIP_CHECK(hipStreamSynchronize(ctx->aux_stream));
        range_set_free(corpus->range_sets[lru]);
        corpus->range_sets.erase(corpus->range_sets.begin() + (long)lru);
    }
    RangeSet *rs = new RangeSet();
    if (hipMalloc(reinterpret_cast<void **>(&rs->dev), head + b_tile + b_chunk + 64) != hipSuccess) {
        (void)hipGetLastError();   // no room for a kept set: the call goes on without one
        delete rs;
        return SMT_OK;
    }
    rs->h1 = h1; rs->h2 = h2; rs->n_in = n_in; rs->nr = nr;
    rs->host_ranges.assign(ranges_in, ranges_in + n_in);
    rs->n_virtual = n_virtual; rs->n_chunks = n_chunks; rs->n_vtiles = n_vtiles;
    rs->d_r = reinterpret_cast<smt_range *>(rs->dev);
    rs->d_p = reinterpret_cast<uint64_t *>(rs->dev + r_bytes);
    rs->d_cp = rs->d_p + (nr + 1);
    rs->d_tp = rs->d_cp + (nr + 1);
    rs->d_tile_table = keep_tables ? reinterpret_cast<uint64_t *>(rs->dev + head) : nullptr;
    rs->d_chunk_table = keep_tables ? reinterpret_cast<uint64_t *>(rs->dev + head + b_tile) : nullptr;
    int rc = ensure_pinned_in(ctx, head);
    if (!rc) {
        // (h_pinned_in is reused by the caller for the queries: the upload must have left it before this returns -- once per set)
        char *pin = reinterpret_cast<char *>(ctx->h_pinned_in);
        memcpy(pin, rr.data(), (size_t)nr * sizeof(smt_range));
        memcpy(pin + r_bytes, prefixes.data(), 3 * p_bytes);
        hipError_t e = hipMemcpyAsync(rs->dev, pin, head, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { set_error("range set upload: %s", hipGetErrorString(e)); rc = SMT_E_HIP; }
    }
    if (rc) { range_set_free(rs); return rc; }
    rs->last_use = ++corpus->range_clock;
    ++corpus->range_set_builds;
    corpus->range_sets.push_back(rs);
    *out = rs;
    return SMT_OK;
}

int range_tile_table(smt_ctx *ctx, const ScanArgs &a, uint64_t *scratch_table, const uint64_t **table)
{
    RangeSet *rs = a.range_set;
    if (rs && rs->d_tile_table) {
        if (!rs->have_tile_table) {
            int rc = launch_build_tile_table(ctx, a.ranges, a.range_tile_prefix, a.n_ranges, a.n_vtiles, rs->d_tile_table);
            if (rc) return rc;
            rs->have_tile_table = true;
        }
        *table = rs->d_tile_table;
        return SMT_OK;
    }
    *table = scratch_table;
    return launch_build_tile_table(ctx, a.ranges, a.range_tile_prefix, a.n_ranges, a.n_vtiles, scratch_table);
}

int range_chunk_table(smt_ctx *ctx, const ScanArgs &a, uint64_t *scratch_table, const uint64_t **table)
{
    RangeSet *rs = a.range_set;
    if (rs && rs->d_chunk_table) {
        if (!rs->have_chunk_table) {
            int rc = launch_build_chunk_table(ctx, a.ranges, a.range_chunk_prefix, a.n_ranges, a.n_chunks, rs->d_chunk_table);
            if (rc) return rc;
            rs->have_chunk_table = true;
        }
        *table = rs->d_chunk_table;
        return SMT_OK;
    }
    *table = scratch_table;
    return launch_build_chunk_table(ctx, a.ranges, a.range_chunk_prefix, a.n_ranges, a.n_chunks, scratch_table);
}

// Exhaustive answer for ONE query whose f32 nomination failed its exactness certificate: K4 collects every row
// whose exact distance is <= bound (its own f32 prefilter carries an 8e-6 guard band; rescoring is exact f64), in
// (distance asc, row asc) order; the answer is the first k_eff of them (after the workspace score filter).
// `bound` is the k-th exact distance found so far -- an upper bound of the true k-th -- or, when fewer than k rows
// passed the workspace threshold, the largest distance that threshold admits.  O(rows <= bound): a cluster of
// near-duplicates costs its own size, exactly what the reference pays for every query (it sorts all N).
static int exact_fallback(smt_ctx *ctx, smt_corpus *corpus, const float *query_dev, const smt_range *ranges_dev,
                          const uint64_t *chunk_prefix_dev, uint32_t nr, uint64_t n_virtual, uint64_t n_chunks, double bound,
                          uint32_t k_eff, bool ws_thr, float thr_score, uint64_t row_base, LocalHits &out)
{
    ThresholdQuery t;
    t.corpus = corpus->d_rows;
    t.rows = corpus->rows;
    t.query = query_dev;
    t.ranges = ranges_dev;
    t.range_chunk_prefix = chunk_prefix_dev;
    t.n_chunks = n_chunks;
    t.n_ranges = nr;
    t.n_virtual = n_virtual;
    t.max_distance = std::nextafter(bound, std::numeric_limits<double>::infinity());  // K4 keeps d < max_distance: include == bound
    const uint32_t *h_rows = nullptr;
    const double *h_dist = nullptr;
    uint64_t n_ok = 0;
    int rc = run_threshold_query(ctx, t, &h_rows, &h_dist, &n_ok);
    if (rc) return rc;
    out.rows.clear();
    out.dist.clear();
    for (uint64_t i = 0; i < n_ok && out.rows.size() < k_eff; ++i) {
        if (ws_thr && !((1.0 - h_dist[i]) > (double)thr_score)) continue;  // store.rs:502-503
        out.rows.push_back(row_base + h_rows[i]);
        out.dist.push_back(h_dist[i]);
    }
    return SMT_OK;
}

// The same re-answer for MANY queries of one call at once.  One sweep of the batched kernel collects, per query, every
// row whose nominating distance is <= bound + F32_ERR_BF16X3 (a superset of the rows with exact distance <= bound);
// they are re-scored exactly, ordered (distance, row) and cut at k -- what exact_fallback does with one K4 scan per
// query.  Queries whose band holds more rows than a candidate buffer (2048) come back in `left` for the K4 route.
// The same sweep answers threshold searches of several queries at once (`strict`: distance < bound, every hit: k_eff = all).
static int batched_fallback(smt_ctx *ctx, smt_corpus *corpus, const float *queries_dev, const std::vector<uint32_t> &redo,
                            const std::vector<double> &bounds, uint64_t k_eff, bool ws_thr, float thr_score, uint64_t row_base,
                            std::vector<LocalHits> &out, std::vector<uint32_t> &left, bool strict = false)
{
    const uint32_t n = (uint32_t)redo.size();
    float *d_qc = nullptr;   // compact copies of the uncertain queries + their f32 thresholds (rare path: plain hipMalloc)
    SMT_HIP_CHECK(hipMalloc(&d_qc, (size_t)n * (SMT_DIM + 2) * sizeof(float)));
    struct Free { float *p; ~Free() { (void)hipFree(p); } } guard{d_qc};
    float *d_tau = d_qc + (size_t)n * SMT_DIM;
    uint32_t *d_redo = reinterpret_cast<uint32_t *>(d_tau + n);
    std::vector<float> tau(n);
    // the sweep runs over the corpus' fp16 operand image when it has one (f16 x 2, half the bytes), else over the f32 rows (bf16 x 3)
    const void *image = nullptr;
    const uint32_t *image_zero = nullptr;
    if (ctx->tune.gemm_image != 0 && corpus->rows <= (1ull << 28)) {
        if (int rc_img = corpus_image_sync(corpus, n, &image, &image_zero)) return rc_img;
    }
    const double band = image ? F32_ERR_F16X2 : F32_ERR_BF16X3;
    for (uint32_t i = 0; i < n; ++i) tau[i] = std::nextafter((float)(bounds[i] + band), std::numeric_limits<float>::infinity());
    SMT_HIP_CHECK(hipMemcpyAsync(d_redo, redo.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    int rc_g = launch_gather_rows256(ctx, queries_dev, d_redo, n, d_qc);   // (a copy per query cost 5 us apiece)
    if (rc_g) return rc_g;
    SMT_HIP_CHECK(hipMemcpyAsync(d_tau, tau.data(), n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    const key_t64 *d_cand = nullptr;
    const unsigned int *d_cnt = nullptr;
    uint32_t stride = 0;
    int rc = launch_gemm_threshold(ctx, corpus->d_rows, corpus->rows, image, image_zero, d_qc, n, d_tau, &d_cand, &d_cnt, &stride);
    if (rc) return rc;
    std::vector<unsigned int> cnt(n);
    std::vector<key_t64> keys((size_t)n * stride);
    SMT_HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt, n * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    SMT_HIP_CHECK(hipMemcpyAsync(keys.data(), d_cand, keys.size() * sizeof(key_t64), hipMemcpyDeviceToHost, ctx->stream));
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // exact distances of every collected row (the scratch that held the candidates is free again)
    std::vector<uint32_t> rows, qidx;
    std::vector<uint64_t> first(n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) {
        first[i] = rows.size();
        if (cnt[i] > stride) continue;   // overflow: K4
        for (uint32_t c = 0; c < cnt[i]; ++c) {
            rows.push_back((uint32_t)(keys[(size_t)i * stride + c] & 0xFFFFFFFFull));
            qidx.push_back(i);
        }
    }
    first[n] = rows.size();
    std::vector<double> dist(rows.size());
    if (!rows.empty()) {   // ONE launch for the rows of all queries (a launch per query cost 11 us apiece: 11.5 of 17.5 ms at 1024 queries)
        const size_t b_rows = (rows.size() * sizeof(uint32_t) + 255) & ~(size_t)255;
        if ((rc = ensure_scratch(ctx, 2 * b_rows + rows.size() * sizeof(double)))) return rc;
        uint32_t *d_rows = reinterpret_cast<uint32_t *>(ctx->d_scratch);
        uint32_t *d_qidx = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(ctx->d_scratch) + b_rows);
        double *d_dist = reinterpret_cast<double *>(reinterpret_cast<char *>(ctx->d_scratch) + 2 * b_rows);
        SMT_HIP_CHECK(hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        SMT_HIP_CHECK(hipMemcpyAsync(d_qidx, qidx.data(), qidx.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        if ((rc = launch_rescore_rows_multi(ctx, corpus->d_rows, d_qc, d_rows, d_qidx, rows.size(), d_dist))) return rc;
        SMT_HIP_CHECK(hipMemcpyAsync(dist.data(), d_dist, rows.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    for (uint32_t i = 0; i < n; ++i) {
        if (cnt[i] > stride) { left.push_back(redo[i]); continue; }
        std::vector<uint64_t> order;
        for (uint64_t c = first[i]; c < first[i + 1]; ++c) {
            if (strict ? !(dist[c] < bounds[i]) : !(dist[c] <= bounds[i])) continue;   // (also drops NaN)
            if (ws_thr && !((1.0 - dist[c]) > (double)thr_score)) continue;         // store.rs:502-503
            order.push_back(c);
        }
        std::sort(order.begin(), order.end(), [&](uint64_t x, uint64_t y) {
            if (dist[x] != dist[y]) return dist[x] < dist[y];
            return rows[x] < rows[y];
        });
        LocalHits &o = out[redo[i]];
        o.rows.clear();
        o.dist.clear();
        for (uint64_t c : order) {
            if (o.rows.size() >= k_eff) break;
            o.rows.push_back(row_base + rows[c]);
            o.dist.push_back(dist[c]);
        }
    }
    return SMT_OK;
}

// The body of smt_search with per-query result vectors instead of caller arrays: group.cpp runs it once per
// local shard (threshold mode / top_k > 64, whose result sizes are not known up front) and exchanges the lists.

// K2 (scan, <= 4 queries per corpus pass) or K3 (batched, one pass for the whole batch)?  8+ queries always take K3.
// With the bf16 x 3 row-register kernel a batch costs about 1.15 single-query passes whatever its size, while a K2 pass
// slows down with every query it carries (per row and query a DPP reduction tree: 10 M rows, 1 / 2 / 4 queries per
// pass = 1.43 / 2.0 / 3.2 ms).  Measured on MI355X, wall ms per call, K2 | K3:
//   10 M rows: 2 queries 2.03 | 1.77, 3: 3.54 | 1.84, 4: 3.24 | 1.71, 5: 4.83 | 1.75, 7: 6.71 | 1.77
//    2 M rows: 2: 0.49 | 0.52, 3: 0.87 | 0.52, 4: 0.75 | 0.53;   1 M rows: 2: 0.27 | 0.36, 3: 0.44 | 0.37, 4: 0.42 | 0.37
//  300 k rows: 3: 0.17 | 0.25 (K3's fixed cost: five level launches + selects)
// => (rounds 2-3) K3 from gemm_min_nq (3) queries on shards of gemm_min_rows_small (1 M) rows, from 2 queries on 4 x that;
// round 4: from gemm_min_nq queries when rows x queries >= 1.2 x gemm_min_rows_small (see below).
static int topk_dispatch(smt_ctx *ctx, smt_corpus *corpus, ScanArgs &a)
{
    // (range-filtered calls whose rows fill their 32-row tiles well enough take the same kernel over a tile table -- tiles_dense,
    // common.h -- and count with the rows they SCAN)
    const bool fast_k3 = ctx->tune.gemm_bf16x3 && ctx->tune.gemm_rowreg && (a.n_ranges == 0 || tiles_dense(a.n_virtual, a.n_vtiles));
    const uint64_t small = (uint64_t)ctx->tune.gemm_min_rows_small;
    const uint64_t scanned = a.n_virtual;
    // A corpus that HAS its fp16 operand image answers even one or two queries through the batched kernel once the shard is
    // large (tuning key image_scan_min_rows; 4 M in round 3, 1.5 M now): one pass over 512-B rows plus the levels and selects beats a scan
    // pass over 1 KiB rows -- 10 M rows: 0.95 against 1.5 ms; the scan kernel keeps the small shards and the async mode.
    // (A resident host asking one query at a time -- `semtools serve` -- never sends the batch of 8 that builds the image of an
    // owned corpus: the fourth small search of a shard this large builds it.)
    const bool whole = corpus->d_rows == a.corpus && corpus->rows == a.rows;
    // Round 4, measured again with the bootstrap plan (profiles/r04_image_scan_sweep.json, us per call, scan of the f32 rows | batched
    // kernel over the image): ONE query 1 M rows 170 | 170, 1.5 M 245 | 214, 2 M 321 | 264, 4 M 612 | 461; TWO queries 0.5 M 134 | 120,
    // 1 M 230 | 167, 2 M 424 | 261 -- the image answers one query from image_scan_min_rows (1.5 M) rows, two and more from a third of that.
    // Late in round 4 the scan kernel reduces the four rows of a chunk together when it carries several queries (scan_kernels.hip
    // SMT_REDUCE_CHUNK4): two queries cost 1.06 x one query's pass, four 1.4 x (were 1.4 x / 2.3 x) -- 1 M rows: 172 / 230 us, 2 M: 315 /
    // 381 -- so the image takes over where it does for one query (two queries) or at two thirds of that (three, four).
    const uint64_t image_min = (uint64_t)ctx->tune.image_scan_min_rows * (a.nq >= 3 ? 2 : 3) / 3;
    const bool scan_sized = fast_k3 && ctx->tune.gemm_image && !a.allow_async && whole &&
                            ctx->tune.image_scan_min_rows > 0 && scanned >= image_min;
    // Round 5, late: every route timed alone across sizes (profiles/r05_sweep_crossover.json, tools/sweep_crossover.py; us per host
    // call, scan kernel | batched kernel over the image): ONE query 500 k rows 110 | 114, 700 k 139 | 132, 1 M 181 | 155, 1.5 M 254 | 193;
    // TWO 200 k 91 | 92, 300 k 103 | 95, 500 k 130 | 111, 1 M 200 | 156; THREE 50 k 88 | 66, 100 k 104 | 88, 1 M 259 | 161; four and more:
    // the image at every size (1000 rows: 80 | 70).  The batched call lost its host read-back in this round and with it ~25 us: an
    // image that EXISTS is used from image_use_min_rows rows by one query, 1/5 of that by two, 1/60 by three and more (unfiltered
    // calls; building one for a corpus that has none keeps the thresholds above).  Measured once more after the bootstrap stride of
    // mid-sized corpora was graded (gemm_topk.hip; same table): ONE query 300 k rows 83 | 86, 500 k 111 | 105; TWO 50 k 68 | 68, 100 k
    // 77 | 71, 300 k 104 | 86 -- image_use_min_rows = 400 k.
    const uint64_t use_min = (uint64_t)ctx->tune.image_use_min_rows * (a.nq >= 3 ? 1 : a.nq == 2 ? 12 : 60) / 60;
    const bool use_sized = fast_k3 && ctx->tune.gemm_image && !a.allow_async && whole && a.n_ranges == 0 && ctx->tune.image_scan_min_rows > 0 &&
                           ctx->tune.image_use_min_rows > 0 && scanned >= use_min;
    if (scan_sized && a.nq < 8 && !corpus->image && corpus->owned && corpus->image_mode == 0 && ctx->tune.corpus_image != 0 &&
        ++corpus->small_searches >= 4) {
        const void *img;
        const uint32_t *zero;
        corpus->image_mode = 1;
        if (int rc_img = corpus_image_sync(corpus, a.nq, &img, &zero)) return rc_img;
        if (corpus->image_mode == 1) corpus->image_mode = 0;   // (-1 when there was no room)
    }
    const bool image_scan = (scan_sized || use_sized) && corpus->image && corpus->image_mode >= 0;
    // Round 4 (bootstrap level: three launches and three select passes fewer per batch) moved the crossover down; measured again
    // (profiles/r04_k2_k3_small.json, us per device-resident call, K2 | K3, no image): 400 k rows 3 queries 175 | 172, 4: 182 | 170,
    // 5: 244 | 171; 200 k rows 4: 115 | 134, 5: 152 | 133, 7: 210 | 138; 100 k rows 5: 104 | 120, 7: 141 | 129; 1 M rows 2: 236 | 275,
    // 3: 377 | 296; 2 M rows 2: 428 | 457 -- K3 from gemm_min_nq queries once rows x queries reaches 1.2 x gemm_min_rows_small.
    // With the chunk-wise reduction (us per call, K2 | K3, profiles/r04_k2_k3_small.json refreshed): 1 M rows 2 queries 172 | 264, 4: 230 |
    // 270, 5: 379 | 280; 400 k rows 4: 134 | 161, 5: 196 | 162; 100 k rows 5: 98 | 111, 7: 133 | 117; 2 M rows 4: 381 | 455, 5: 672 | 475 --
    // up to four queries stay on the scan kernel (gemm_min_nq = 5), five to seven move as before.
    // Same sweep, scan kernel | batched kernel over f32 rows: FIVE queries 1000 rows 89 | 76, 100 k 131 | 99, 2 M 709 | 472 -- the batched
    // kernel at every size (unfiltered; the rows x queries bound above stays for document subsets, whose tile table is one more
    // launch); THREE 50 k 88 | 79, 300 k 146 | 129, 700 k 213 | 208, 1 M 259 | 266; FOUR 50 k 95 | 80, 300 k 152 | 129, 1 M 266 | 261, 1.5 M
    // 340 | 360; TWO: the scan kernel everywhere (2 M: 345 | 443).  So up to two queries below gemm_min_nq (three, four) take the
    // batched kernel in the band [small / 50, 0.8 small] = 20 k .. 800 k rows.
    const bool mid_band = a.n_ranges == 0 && a.nq >= 3 && a.nq + 2 >= (uint32_t)ctx->tune.gemm_min_nq && small > 0 &&
                          scanned >= small / 50 && scanned <= small / 5 * 4;
    const bool batched = a.nq >= 8 ||
                         (fast_k3 && a.nq >= (uint32_t)ctx->tune.gemm_min_nq && (a.n_ranges == 0 || scanned * a.nq >= small + small / 5)) ||
                         (fast_k3 && mid_band) || image_scan;
    if (batched && fast_k3 && whole) {
        if (int rc_img = corpus_image_sync(corpus, a.nq, &a.image, &a.image_zero)) return rc_img;
    }
    int rc = batched ? launch_gemm_topk(ctx, a) : launch_scan_topk(ctx, a);
    if (rc == SMT_E_UNSUPPORTED && batched) rc = launch_scan_topk(ctx, a);
    return rc;
}

// Store::search_line_embeddings with a ZERO query vector (an empty query, or one made of unknown tokens only: model2vec pools it to
// zeros).  qdrant's cosine_preprocess leaves a vector with |x|^2 < f32::EPSILON as it is, so the query scores 0 against EVERY point --
// distance 1.0 for all of them, zero rows included -- where simsimd's rule for search_documents says (zero, zero) -> distance 0
// (oracle: orc_search_line_embeddings against orc_cosine_*; src/workspace/store.rs:500-531).  The answer is a constant, so it is
// written here instead of being computed: with a threshold nothing unless 0 > 1 - max_distance (f32), else the first top_k rows of
// the subset in storage order (equal scores: earlier row first, as in the oracle), each at distance 1.0.
bool query_is_zero(const float *q)
{
    for (uint32_t d = 0; d < SMT_DIM; ++d)
        if (q[d] != 0.0f) return false;
    return true;
}

void workspace_zero_query_hits(const smt_range *ranges, uint32_t n_ranges, uint64_t n_rows, uint32_t top_k, bool has_thr, double max_distance,
                               uint64_t row_base, LocalHits &out)
{
    out.rows.clear();
    out.dist.clear();
    if (has_thr && !(0.0f > 1.0f - (float)max_distance)) return;
    auto take = [&](uint64_t b, uint64_t e) {
        for (uint64_t r = b; r < e && out.rows.size() < top_k; ++r) { out.rows.push_back(row_base + r); out.dist.push_back(1.0); }
    };
    if (n_ranges == 0) take(0, n_rows);
    for (uint32_t i = 0; i < n_ranges && out.rows.size() < top_k; ++i) take(ranges[i].begin, ranges[i].end);
}

static int search_local_host_impl(smt_corpus *corpus, const float *queries, uint32_t nq, uint32_t top_k, double max_distance, int mode,
                                  const smt_range *ranges, uint32_t n_ranges, uint64_t row_base, std::vector<LocalHits> &out);

int search_local_host(smt_corpus *corpus, const float *queries, uint32_t nq, uint32_t top_k, double max_distance, int mode,
                      const smt_range *ranges, uint32_t n_ranges, uint64_t row_base, std::vector<LocalHits> &out)
{
    if (mode != SMT_MODE_WORKSPACE || !queries || nq == 0)
        return search_local_host_impl(corpus, queries, nq, top_k, max_distance, mode, ranges, n_ranges, row_base, out);
    // workspace mode: zero queries never reach the GPU (every row ties at distance 1.0: the kernels would flag the answer uncertain
    // and the exhaustive re-answer would collect the whole subset to order a constant)
    std::vector<uint32_t> live;
    for (uint32_t q = 0; q < nq; ++q)
        if (!query_is_zero(queries + (size_t)q * SMT_DIM)) live.push_back(q);
    if (live.size() == nq) return search_local_host_impl(corpus, queries, nq, top_k, max_distance, mode, ranges, n_ranges, row_base, out);
    SMT_REQUIRE(corpus != nullptr, "corpus");
    std::vector<float> packed(live.size() * (size_t)SMT_DIM);
    for (size_t j = 0; j < live.size(); ++j) memcpy(&packed[j * SMT_DIM], queries + (size_t)live[j] * SMT_DIM, SMT_DIM * sizeof(float));
    std::vector<LocalHits> part;
    // (an all-zero batch still goes through the argument checks of the search proper: ranges, mode, the corpus' device)
    int rc = search_local_host_impl(corpus, packed.empty() ? queries : packed.data(), packed.empty() ? 0u : (uint32_t)live.size(), top_k,
                                    max_distance, mode, ranges, n_ranges, row_base, part);
    if (rc) return rc;
    if (n_ranges) {   // (validated by the search proper only when it had a query to answer)
        uint64_t prev_end = 0;
        for (uint32_t i = 0; i < n_ranges; ++i) {
            SMT_REQUIRE(ranges[i].begin <= ranges[i].end && ranges[i].end <= corpus->rows && (i == 0 || ranges[i].begin >= prev_end),
                        "ranges must be sorted, disjoint and inside the corpus");
            prev_end = ranges[i].end;
        }
    }
    out.assign(nq, LocalHits());
    for (size_t j = 0; j < live.size(); ++j) out[live[j]] = std::move(part[j]);
    for (uint32_t q = 0; q < nq; ++q)
        if (query_is_zero(queries + (size_t)q * SMT_DIM))
            workspace_zero_query_hits(ranges, n_ranges, corpus->rows, top_k, !std::isnan(max_distance), max_distance, row_base, out[q]);
    return SMT_OK;
}

static int search_local_host_impl(smt_corpus *corpus, const float *queries, uint32_t nq, uint32_t top_k, double max_distance, int mode,
                                  const smt_range *ranges, uint32_t n_ranges, uint64_t row_base, std::vector<LocalHits> &out)
{
    SMT_REQUIRE(corpus != nullptr, "corpus");
    SMT_REQUIRE(mode == SMT_MODE_DOCUMENTS || mode == SMT_MODE_WORKSPACE, "mode");
    SMT_REQUIRE(nq == 0 || queries, "null argument");
    SMT_REQUIRE(n_ranges == 0 || ranges != nullptr, "ranges");
    smt_ctx *ctx = corpus->ctx;
    int rc = bind_device(ctx);
    if (rc) return rc;
    out.assign(nq, LocalHits());
    if (nq == 0) return SMT_OK;
    if ((rc = require_queries_domain_host(queries, nq, "search"))) return rc;   // (domain.hip: finite, ordinary magnitudes)

    const bool has_thr = !std::isnan(max_distance);
    const bool all_under_threshold = (mode == SMT_MODE_DOCUMENTS) && has_thr;

    // drop empty ranges; total rows to scan.  A list searched before has its device copy (and tables) kept on the corpus.
    std::vector<smt_range> rr;
    uint64_t n_virtual = corpus->rows;
    RangeSet *rset = nullptr;
    bool rset_build = false;
    uint64_t rh1 = 0, rh2 = 0;
    if (n_ranges) {
        uint64_t total = 0;
        if ((rc = range_set_find(corpus, ranges, n_ranges, &total, &rset, &rset_build, &rh1, &rh2))) return rc;
        n_virtual = total;
        if (!rset)
            for (uint32_t i = 0; i < n_ranges; ++i) if (ranges[i].end > ranges[i].begin) rr.push_back(ranges[i]);
    }
    if (n_virtual == 0) return SMT_OK;
    if (!all_under_threshold && top_k == 0) return SMT_OK;  // take(0) / store.rs:489-491
    std::vector<uint64_t> prefixes;
    if (!rset && !rr.empty()) {
        range_prefixes(rr, prefixes);
        if (rset_build && (rc = range_set_build(corpus, ranges, n_ranges, rh1, rh2, rr, prefixes, n_virtual, &rset))) return rc;
    }

    // ---- device staging: queries, ranges(+prefix)
    const uint32_t nr = rset ? rset->nr : (uint32_t)rr.size();
    const size_t q_bytes = (size_t)nq * SMT_DIM * sizeof(float);
    const size_t r_bytes = (size_t)nr * sizeof(smt_range);
    const size_t p_bytes = (size_t)(nr + 1) * sizeof(uint64_t);
    // one persistent staging buffer per context: [queries | ranges | 3 prefixes | result lists] (no per-call hipMalloc/hipFree)
    const size_t in_bytes = (q_bytes + r_bytes + 3 * p_bytes + 255) & ~(size_t)255;
    const uint32_t k_stage = all_under_threshold ? 0u : (uint32_t)std::min<uint64_t>(std::min<uint64_t>(top_k, n_virtual), 64);
    const size_t out_bytes_stage = (size_t)nq * k_stage * 16 + (size_t)2 * nq * sizeof(uint64_t);
    if ((rc = ensure_stage(ctx, in_bytes + out_bytes_stage + 64))) return rc;
    char *stage = reinterpret_cast<char *>(ctx->d_stage);
    float *d_q = reinterpret_cast<float *>(stage);
    smt_range *d_r = reinterpret_cast<smt_range *>(stage + q_bytes);
    uint64_t *d_p = reinterpret_cast<uint64_t *>(stage + q_bytes + r_bytes);
    uint64_t *d_cp = d_p + (nr + 1), *d_tp = d_cp + (nr + 1);
    // queries, ranges and the three prefixes are assembled in ONE pinned buffer (laid out like the device stage) and go up in one
    // copy: four pageable hipMemcpyAsync calls -- each staged by the runtime before it returns -- were ~50 us of a 0.6 ms call
    const size_t up_bytes = q_bytes + (nr && !rset ? r_bytes + 3 * p_bytes : 0);
    if ((rc = ensure_pinned_in(ctx, up_bytes))) return rc;
    {
        char *pin = reinterpret_cast<char *>(ctx->h_pinned_in);
        memcpy(pin, queries, q_bytes);
        if (nr && !rset) {
            memcpy(pin + q_bytes, rr.data(), r_bytes);
            memcpy(pin + q_bytes + r_bytes, prefixes.data(), 3 * p_bytes);
        }
        SMT_HIP_CHECK(hipMemcpyAsync(stage, pin, up_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    if (rset) { d_r = rset->d_r; d_p = rset->d_p; d_cp = rset->d_cp; d_tp = rset->d_tp; }   // (only the queries went up)
    const uint64_t n_chunks = rset ? rset->n_chunks : nr ? prefixes[2 * (nr + 1) - 1] : 0;
    const uint64_t n_vtiles = rset ? rset->n_vtiles : nr ? prefixes[3 * (nr + 1) - 1] : 0;

    if (!all_under_threshold) {
        // ---------------- top-k (optionally with the workspace score threshold)
        const uint32_t k_eff = (uint32_t)std::min<uint64_t>(top_k, n_virtual);
        if (k_eff > SCAN_MAX_K) {
            // large-k request (also k in 57..64, where the f32 scan's candidate lists have no room left for the
            // guard band): all keys + sort + exact rescoring of k + guard candidates
            const uint64_t guard = std::max<uint64_t>(64, k_eff / 16);
            const uint64_t n_cand = std::min<uint64_t>(n_virtual, (uint64_t)k_eff + guard);
            const bool ws_thr = (mode == SMT_MODE_WORKSPACE && has_thr);
            const float thr_score = 1.0f - (float)max_distance;
            for (uint32_t q = 0; q < nq; ++q) {
                std::vector<uint32_t> c_rows;
                std::vector<double> c_dist;
                float next_d32 = 0.f;
                rc = launch_largek_candidates(ctx, corpus->d_rows, d_q + (size_t)q * SMT_DIM, nr ? d_r : nullptr,
                                              nr ? d_p : nullptr, nr, n_virtual, n_cand, c_rows, c_dist, &next_d32);
                if (rc) return rc;
                std::vector<uint64_t> order;
                for (uint64_t i = 0; i < c_rows.size(); ++i) {
                    if (c_dist[i] != c_dist[i]) continue;                                   // NaN rows never match
                    if (ws_thr && !((1.0 - c_dist[i]) > (double)thr_score)) continue;      // store.rs:502-503
                    order.push_back(i);
                }
                std::sort(order.begin(), order.end(), [&](uint64_t x, uint64_t y) {
                    if (c_dist[x] != c_dist[y]) return c_dist[x] < c_dist[y];
                    return c_rows[x] < c_rows[y];
                });
                const uint64_t n = std::min<uint64_t>(order.size(), k_eff);
                out[q].rows.resize(n);
                out[q].dist.resize(n);
                for (uint64_t i = 0; i < n; ++i) {
                    out[q].rows[i] = row_base + c_rows[order[i]];
                    out[q].dist[i] = c_dist[order[i]];
                }
                // exactness certificate (SelectArgs::f32_err): rows outside the candidates have exact distance >= floor_out
                const double floor_out = (double)next_d32 - F32_ERR_SCAN;
                const bool certain = n == k_eff ? floor_out > out[q].dist[n - 1]
                                                : (ws_thr ? !((1.0 - floor_out) > (double)thr_score) : next_d32 == __builtin_inff());
                if (!certain) {
                    const double bound = n == k_eff ? out[q].dist[n - 1] : 1.0 - (double)thr_score;
                    rc = exact_fallback(ctx, corpus, d_q + (size_t)q * SMT_DIM, nr ? d_r : nullptr, nr ? d_cp : nullptr, nr,
                                        n_virtual, n_chunks, bound, k_eff, ws_thr, thr_score, row_base, out[q]);
                    if (rc) return rc;
                }
            }
            return SMT_OK;
        }
        const size_t o_rows = (size_t)nq * k_eff * sizeof(uint64_t);
        const size_t o_dist = (size_t)nq * k_eff * sizeof(double);
        const size_t o_cnt = (size_t)2 * nq * sizeof(uint64_t);  // counts, then the "uncertain" flags
        char *outs = stage + in_bytes;
        uint64_t *d_orow = reinterpret_cast<uint64_t *>(outs);
        double *d_odist = reinterpret_cast<double *>(outs + o_rows);
        uint64_t *d_ocnt = reinterpret_cast<uint64_t *>(outs + o_rows + o_dist);

        ScanArgs a;
        a.corpus = corpus->d_rows;
        a.rows = corpus->rows;
        a.queries = d_q;
        a.nq = nq;
        a.k_out = k_eff;
        a.ranges = nr ? d_r : nullptr;
        a.range_prefix = nr ? d_p : nullptr;
        a.range_chunk_prefix = nr ? d_cp : nullptr;
        a.n_chunks = n_chunks;
        a.range_tile_prefix = nr ? d_tp : nullptr;
        a.n_vtiles = n_vtiles;
        a.n_ranges = nr;
        a.n_virtual = n_virtual;
        a.ws_threshold = (mode == SMT_MODE_WORKSPACE && has_thr) ? 1 : 0;
        a.ws_thr_score = 1.0f - (float)max_distance;  // store.rs:502-503
        a.row_base = row_base;
        a.out_rows = d_orow;
        a.out_dist = d_odist;
        a.out_counts = d_ocnt;
        a.out_uncertain = d_ocnt + nq;
        a.range_set = rset;
        if ((rc = ensure_pinned(ctx, 64 + o_rows + o_dist + o_cnt))) return rc;
        // the answers' place in the pinned buffer: behind a 64-byte line whose first word is the completion word of a delivered answer
        char *h_ans = reinterpret_cast<char *>(ctx->h_pinned) + 64;
        // A SMALL answer is delivered by the select kernel itself (common.h Delivery): its last block copies the device block
        // [rows | distances | counts | flags] into the pinned buffer and stores this call's sequence number behind it; the host waits
        // for that word -- no D2H copy command, no hipStreamSynchronize: ~10 us of every small call (profiles/r06_call_floor.json,
        // profiles/r06_small_calls.json).  One select launch answers the whole call (<= 32 queries: below the batched kernel's pass size).
        const bool direct = ctx->tune.direct_delivery != 0 && nq <= 32 && o_rows + o_dist + o_cnt <= 8192;
        Delivery dl;
        volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(ctx->h_pinned);
        if (direct) {
            dl.dev_out = reinterpret_cast<const unsigned long long *>(outs);
            dl.host_out = reinterpret_cast<unsigned long long *>(h_ans);
            dl.n_words = (uint32_t)((o_rows + o_dist + o_cnt) / 8);
            dl.host_flag = const_cast<unsigned long long *>(flag);
            dl.seq = ++ctx->deliver_seq;
            dl.done = ctx->d_status + 4;
            *flag = 0;
            a.deliver = &dl;
        }
        // K2 or K3: topk_dispatch above
        rc = topk_dispatch(ctx, corpus, a);
        if (rc) return rc;
        if (direct) {
            // spin on the completion word (a small search is over in tens of microseconds); past 200 us ask the runtime, which also
            // reports a launch that failed
            const auto t0 = std::chrono::steady_clock::now();
            unsigned long long got = 0;
            for (unsigned spins = 1; (got = *flag) == 0; ++spins) {
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();   // (a polite spin: the sibling hyperthread keeps its issue slots)
#endif
                if ((spins & 63) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
                    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    got = *flag;
                    break;
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            if (got != dl.seq) {
                (void)hipStreamSynchronize(ctx->stream);   // (whatever is left of the launch) -- and the block counter starts from zero again
                (void)hipMemsetAsync(ctx->d_status + 4, 0, sizeof(unsigned long long), ctx->stream);
                set_error("the select kernel did not deliver its answer (completion word %llu, expected %llu)", got, dl.seq);
                return SMT_E_HIP;
            }
            ++ctx->deliveries;
        } else {
            SMT_HIP_CHECK(hipMemcpyAsync(h_ans, outs, o_rows + o_dist + o_cnt, hipMemcpyDeviceToHost, ctx->stream));
            SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
        const uint64_t *h_rows = reinterpret_cast<const uint64_t *>(h_ans);
        const double *h_dist = reinterpret_cast<const double *>(h_ans + o_rows);
        const uint64_t *h_cnt = reinterpret_cast<const uint64_t *>(h_ans + o_rows + o_dist);
        std::vector<uint32_t> redo;
        for (uint32_t q = 0; q < nq; ++q) {
            const uint64_t n = h_cnt[q];
            out[q].rows.assign(h_rows + (size_t)q * k_eff, h_rows + (size_t)q * k_eff + n);
            out[q].dist.assign(h_dist + (size_t)q * k_eff, h_dist + (size_t)q * k_eff + n);
            if (h_cnt[nq + q]) redo.push_back(q);  // (h_pinned is reused by the fallback: copy everything out first)
        }
        // queries whose f32 nomination could not be proven sufficient (a cluster of near-ties around the k-th
        // place that is wider than the guard band): answer them exhaustively -- several of them with ONE batched
        // threshold pass over the shard (batched_fallback), the rest (and whatever overflows there) one K4 scan each
        const bool ws_thr = a.ws_threshold != 0;
        auto bound_of = [&](uint32_t q) {
            return out[q].rows.size() == k_eff ? out[q].dist.back() : 1.0 - (double)a.ws_thr_score;
        };
        if (nr == 0 && redo.size() >= 2 && ctx->tune.gemm_bf16x3 && ctx->tune.gemm_rowreg &&
            corpus->rows >= (uint64_t)ctx->tune.fallback_batch_min_rows) {
            std::vector<double> bounds;
            for (uint32_t q : redo) bounds.push_back(bound_of(q));
            std::vector<uint32_t> left;
            rc = batched_fallback(ctx, corpus, d_q, redo, bounds, k_eff, ws_thr, a.ws_thr_score, row_base, out, left);
            if (rc) return rc;
            redo.swap(left);
        }
        for (uint32_t q : redo) {
            rc = exact_fallback(ctx, corpus, d_q + (size_t)q * SMT_DIM, nr ? d_r : nullptr, nr ? d_cp : nullptr, nr, n_virtual,
                                n_chunks, bound_of(q), k_eff, ws_thr, a.ws_thr_score, row_base, out[q]);
            if (rc) return rc;
        }
        return SMT_OK;
    }

    // ---------------- all rows with distance < max_distance (mod.rs:88-89,115-116)
    // several queries on a large unfiltered shard: ONE sweep of the batched kernel collects every query's hits (up to a
    // candidate buffer, 2048 rows, each); a query with more hits than that takes the streaming K4 scan below
    std::vector<uint32_t> todo(nq);
    for (uint32_t q = 0; q < nq; ++q) todo[q] = q;
    if (nr == 0 && nq >= 2 && ctx->tune.gemm_bf16x3 && ctx->tune.gemm_rowreg && max_distance <= 2.5 &&
        corpus->rows >= (uint64_t)ctx->tune.fallback_batch_min_rows) {
        std::vector<double> bounds(nq, max_distance);
        std::vector<uint32_t> left;
        rc = batched_fallback(ctx, corpus, d_q, todo, bounds, ~0ull, false, 0.f, row_base, out, left, /*strict=*/true);
        if (rc) return rc;
        todo.swap(left);
    }
    for (uint32_t q : todo) {
        ThresholdQuery t;
        t.corpus = corpus->d_rows;
        t.rows = corpus->rows;
        t.query = d_q + (size_t)q * SMT_DIM;
        t.ranges = nr ? d_r : nullptr;
        t.range_chunk_prefix = nr ? d_cp : nullptr;
        t.n_chunks = n_chunks;
        t.n_ranges = nr;
        t.n_virtual = n_virtual;
        t.max_distance = max_distance;
        const uint32_t *h_rows = nullptr;
        const double *h_dist = nullptr;
        uint64_t n_ok = 0;
        if ((rc = run_threshold_query(ctx, t, &h_rows, &h_dist, &n_ok))) return rc;
        out[q].rows.resize(n_ok);
        for (uint64_t i = 0; i < n_ok; ++i) out[q].rows[i] = row_base + h_rows[i];
        out[q].dist.assign(h_dist, h_dist + n_ok);
    }
    return SMT_OK;
}

// Copy per-query hit lists into the caller's [nq x out_cap] arrays; counts hold the TRUE sizes.
int deliver_hits(const std::vector<LocalHits> &hits, uint64_t *out_rows, double *out_dist, uint64_t *out_counts, uint64_t out_cap)
{
    bool truncated = false;
    for (size_t q = 0; q < hits.size(); ++q) {
        const uint64_t n = hits[q].rows.size();
        out_counts[q] = n;
        const uint64_t w = std::min<uint64_t>(n, out_cap);
        if (n > out_cap) truncated = true;
        if (w) {
            SMT_REQUIRE(out_rows && out_dist, "null output");
            memcpy(out_rows + q * out_cap, hits[q].rows.data(), (size_t)w * sizeof(uint64_t));
            memcpy(out_dist + q * out_cap, hits[q].dist.data(), (size_t)w * sizeof(double));
        }
    }
    if (truncated) { set_error("out_cap smaller than the number of hits"); return SMT_E_TRUNCATED; }
    return SMT_OK;
}

// One shard's top-k with everything on the device (the exchange path of group.cpp).  queries_dev [nq x 256];
// ranges_local = sorted, disjoint LOCAL row ranges (host array); filtered && n_ranges == 0 means "the filter
// leaves this shard nothing to scan".  packed_dev [nq][2][k_pad] receives global rows, then f64 distance bit
// patterns, padded with (UINT64_MAX, +inf).  1 <= k_pad <= SCAN_MAX_K.  Enqueued on the context's stream (the
// select stage on the aux stream when allow_async and the async_select tuning key say so); no host sync.
int search_topk_packed_local(smt_corpus *corpus, const float *queries_dev, uint32_t nq, uint32_t k_pad, int ws_threshold,
                             float ws_thr_score, const smt_range *ranges_local, uint32_t n_ranges, bool filtered,
                             uint64_t row_base, uint64_t *packed_dev, uint64_t *uncertain_dev, bool allow_async)
{
    SMT_REQUIRE(corpus && queries_dev && packed_dev, "null argument");
    SMT_REQUIRE(k_pad >= 1 && k_pad <= SCAN_MAX_K, "top_k of the device exchange path must be in [1, 56]");
    smt_ctx *ctx = corpus->ctx;
    uint64_t n_virtual = corpus->rows;
    std::vector<smt_range> rr;
    RangeSet *rset = nullptr;
    bool rset_build = false;
    uint64_t rh1 = 0, rh2 = 0;
    if (filtered) {
        uint64_t total = 0;
        int rcv = range_set_find(corpus, ranges_local, n_ranges, &total, &rset, &rset_build, &rh1, &rh2);
        if (rcv) return rcv;
        if (!rset)
            for (uint32_t i = 0; i < n_ranges; ++i) if (ranges_local[i].end > ranges_local[i].begin) rr.push_back(ranges_local[i]);
        n_virtual = total;
    }
    const uint32_t k_eff = (uint32_t)std::min<uint64_t>(k_pad, n_virtual);
    const bool async = allow_async && ctx->tune.async_select && nq == 1 && !filtered && k_eff == k_pad;
    int rc = bind_device(ctx, !async);
    if (rc) return rc;
    if (k_eff < k_pad) {  // short or empty shard: padding first, the select then overwrites the head of each list
        if ((rc = launch_merge_topk_packed_on(ctx, ctx->stream, packed_dev, 0, nq, 1, k_pad, packed_dev))) return rc;
        if (k_eff == 0) {
            if (uncertain_dev) SMT_HIP_CHECK(hipMemsetAsync(uncertain_dev, 0, (size_t)nq * sizeof(uint64_t), ctx->stream));
            return SMT_OK;
        }
    }
    std::vector<uint64_t> prefixes;
    if (!rset && !rr.empty()) {
        range_prefixes(rr, prefixes);
        if (rset_build && (rc = range_set_build(corpus, ranges_local, n_ranges, rh1, rh2, rr, prefixes, n_virtual, &rset))) return rc;
    }
    const uint32_t nr = rset ? rset->nr : (uint32_t)rr.size();
    smt_range *d_r = nullptr;
    uint64_t *d_p = nullptr, *d_cp = nullptr, *d_tp = nullptr;
    uint64_t n_chunks = 0, n_vtiles = 0;
    if (rset) {
        d_r = rset->d_r; d_p = rset->d_p; d_cp = rset->d_cp; d_tp = rset->d_tp;
        n_chunks = rset->n_chunks; n_vtiles = rset->n_vtiles;
    } else if (nr) {
        const size_t r_bytes = (size_t)nr * sizeof(smt_range), p_bytes = (size_t)(nr + 1) * sizeof(uint64_t);
        if ((rc = ensure_stage(ctx, r_bytes + 3 * p_bytes + 64))) return rc;
        char *stage = reinterpret_cast<char *>(ctx->d_stage);
        d_r = reinterpret_cast<smt_range *>(stage);
        d_p = reinterpret_cast<uint64_t *>(stage + r_bytes);
        d_cp = d_p + (nr + 1);
        d_tp = d_cp + (nr + 1);
        n_chunks = prefixes[2 * (nr + 1) - 1];
        n_vtiles = prefixes[3 * (nr + 1) - 1];
        SMT_HIP_CHECK(hipMemcpyAsync(d_r, rr.data(), r_bytes, hipMemcpyHostToDevice, ctx->stream));
        SMT_HIP_CHECK(hipMemcpyAsync(d_p, prefixes.data(), 3 * p_bytes, hipMemcpyHostToDevice, ctx->stream));
        SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // the host vectors die with this frame
    }
    ScanArgs a;
    a.corpus = corpus->d_rows;
    a.rows = corpus->rows;
    a.queries = queries_dev;
    a.nq = nq;
    a.k_out = k_eff;
    a.ranges = d_r;
    a.range_prefix = d_p;
    a.range_chunk_prefix = d_cp;
    a.n_chunks = n_chunks;
    a.range_tile_prefix = d_tp;
    a.n_vtiles = n_vtiles;
    a.n_ranges = nr;
    a.n_virtual = n_virtual;
    a.ws_threshold = ws_threshold;
    a.ws_thr_score = ws_thr_score;
    a.row_base = row_base;
    a.out_rows = packed_dev;
    a.out_dist = reinterpret_cast<double *>(packed_dev + k_pad);
    a.out_counts = nullptr;
    a.out_uncertain = uncertain_dev;
    a.allow_async = async;
    a.out_stride = (uint64_t)2 * k_pad;
    a.range_set = rset;
    rc = topk_dispatch(ctx, corpus, a);
    return rc;
}

}  // namespace smt

extern "C" {

int smt_search(smt_corpus *corpus, const float *queries, uint32_t nq, uint32_t top_k, double max_distance, int mode,
               const smt_range *ranges, uint32_t n_ranges, uint64_t row_base, uint64_t *out_rows, double *out_dist,
               uint64_t *out_counts, uint64_t out_cap)
try {
    SMT_REQUIRE(nq == 0 || out_counts, "null argument");
    std::vector<LocalHits> hits;
    int rc = search_local_host(corpus, queries, nq, top_k, max_distance, mode, ranges, n_ranges, row_base, hits);
    if (rc) return rc;
    return deliver_hits(hits, out_rows, out_dist, out_counts, out_cap);
} catch (...) { return smt::api_catch(); }

int smt_debug_deliveries(smt_ctx *ctx, uint64_t *count)
try {
    SMT_REQUIRE(ctx != nullptr && count != nullptr, "null argument");
    *count = ctx->deliveries;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_debug_range_sets(const smt_corpus *corpus, uint64_t *kept, uint64_t *hits, uint64_t *builds)
try {
    SMT_REQUIRE(corpus != nullptr, "corpus");
    if (kept) *kept = corpus->range_sets.size();
    if (hits) *hits = corpus->range_set_hits;
    if (builds) *builds = corpus->range_set_builds;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_search_topk_device(smt_corpus *corpus, const float *queries_dev, uint32_t nq, uint32_t top_k, uint64_t row_base,
                           uint64_t *out_rows_dev, double *out_dist_dev)
try {
    return smt_search_topk_device_ex(corpus, queries_dev, nq, top_k, row_base, out_rows_dev, out_dist_dev, nullptr);
} catch (...) { return smt::api_catch(); }

int smt_search_topk_device_ex(smt_corpus *corpus, const float *queries_dev, uint32_t nq, uint32_t top_k, uint64_t row_base,
                              uint64_t *out_rows_dev, double *out_dist_dev, uint32_t *out_status_dev)
try {
    SMT_REQUIRE(corpus != nullptr, "corpus");
    SMT_REQUIRE(nq == 0 || (queries_dev && out_rows_dev && out_dist_dev), "null argument");
    SMT_REQUIRE(top_k >= 1 && top_k <= SCAN_MAX_K, "top_k must be in [1, 56]");
    smt_ctx *ctx = corpus->ctx;
    const bool async = ctx->tune.async_select && nq == 1 && corpus->rows > 0;  // launch_scan_topk keeps the pipeline going
    int rc = bind_device(ctx, !async);
    if (rc) return rc;
    if (nq == 0) return SMT_OK;
    ScanArgs a;
    a.corpus = corpus->d_rows;
    a.rows = corpus->rows;
    a.queries = queries_dev;
    a.nq = nq;
    a.k_out = top_k;
    a.ranges = nullptr;
    a.range_prefix = nullptr;
    a.range_chunk_prefix = nullptr;
    a.n_chunks = 0;
    a.n_ranges = 0;
    a.n_virtual = corpus->rows;
    a.ws_threshold = 0;
    a.ws_thr_score = 0.f;
    a.row_base = row_base;
    a.out_rows = out_rows_dev;
    a.out_dist = out_dist_dev;
    a.out_counts = nullptr;
    a.out_status = out_status_dev;
    a.allow_async = async;
    if (corpus->rows == 0) {
        // nothing to scan: fill with padding through the merge kernel on zero lists (an empty answer is a proved one)
        if (out_status_dev) SMT_HIP_CHECK(hipMemsetAsync(out_status_dev, 0, (size_t)nq * sizeof(uint32_t), ctx->stream));
        return launch_merge_topk(ctx, out_rows_dev, out_dist_dev, 0, nq, 1, top_k, out_rows_dev, out_dist_dev);
    }
    rc = topk_dispatch(ctx, corpus, a);
    return rc;
} catch (...) { return smt::api_catch(); }

int smt_debug_batched_scores(smt_corpus *corpus, const float *queries, uint32_t nq, uint64_t first_row, uint32_t n_rows,
                             float *out)
try {
    SMT_REQUIRE(corpus && queries && out, "null argument");
    SMT_REQUIRE(first_row + n_rows <= corpus->rows, "row range outside the corpus");
    smt_ctx *ctx = corpus->ctx;
    int rc = bind_device(ctx, true);
    if (rc) return rc;
    const size_t b_q = (size_t)nq * 256 * 4, b_out = (size_t)n_rows * 32 * 4;
    if ((rc = ensure_scratch(ctx, b_q + b_out + 256))) return rc;
    float *d_q = reinterpret_cast<float *>(ctx->d_scratch);
    float *d_out = reinterpret_cast<float *>(reinterpret_cast<char *>(ctx->d_scratch) + ((b_q + 255) & ~(size_t)255));
    SMT_HIP_CHECK(hipMemcpyAsync(d_q, queries, b_q, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_gemm_debug_scores(ctx, corpus->d_rows, first_row, n_rows, d_q, nq, d_out))) return rc;
    SMT_HIP_CHECK(hipMemcpyAsync(out, d_out, b_out, hipMemcpyDeviceToHost, ctx->stream));
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_merge_topk(const uint64_t *rows, const double *dist, uint32_t n_lists, uint32_t nq, uint32_t k_in, uint32_t k_out,
                   uint64_t *out_rows, double *out_dist, uint64_t *out_counts)
try {
    SMT_REQUIRE((rows && dist) || n_lists == 0 || nq == 0 || k_in == 0, "null input");
    SMT_REQUIRE(nq == 0 || k_out == 0 || (out_rows && out_dist), "null output");
    std::vector<std::pair<double, uint64_t>> cand;
    for (uint32_t q = 0; q < nq; ++q) {
        cand.clear();
        for (uint32_t l = 0; l < n_lists; ++l)
            for (uint32_t i = 0; i < k_in; ++i) {
                const size_t idx = ((size_t)l * nq + q) * k_in + i;
                if (rows[idx] != UINT64_MAX) cand.emplace_back(dist[idx], rows[idx]);
            }
        std::sort(cand.begin(), cand.end());
        const uint64_t n = std::min<uint64_t>(cand.size(), k_out);
        for (uint32_t i = 0; i < k_out; ++i) {
            out_rows[(size_t)q * k_out + i] = i < n ? cand[i].second : UINT64_MAX;
            out_dist[(size_t)q * k_out + i] = i < n ? cand[i].first : std::numeric_limits<double>::infinity();
        }
        if (out_counts) out_counts[q] = n;
    }
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_merge_topk_device(smt_ctx *ctx, const uint64_t *rows_dev, const double *dist_dev, uint32_t n_lists, uint32_t nq,
                          uint32_t k_in, uint32_t k_out, uint64_t *out_rows_dev, double *out_dist_dev)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    SMT_REQUIRE(nq == 0 || (out_rows_dev && out_dist_dev), "null output");
    if ((rc = bind_device(ctx))) return rc;
    if (nq == 0 || k_out == 0) return SMT_OK;
    return launch_merge_topk(ctx, rows_dev, dist_dev, n_lists, nq, k_in, k_out, out_rows_dev, out_dist_dev);
} catch (...) { return smt::api_catch(); }

int smt_merge_topk_packed_device(smt_ctx *ctx, const uint64_t *packed_dev, uint32_t n_lists, uint32_t nq, uint32_t k_in,
                                 uint32_t k_out, uint64_t *out_packed_dev)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    SMT_REQUIRE(nq == 0 || k_out == 0 || (packed_dev && out_packed_dev), "null argument");
    if ((rc = bind_device(ctx, !(ctx->tune.merge_on_aux && ctx->aux_stream)))) return rc;
    if (nq == 0 || k_out == 0) return SMT_OK;
    return launch_merge_topk_packed(ctx, packed_dev, n_lists, nq, k_in, k_out, out_packed_dev);
} catch (...) { return smt::api_catch(); }

}  // extern "C"
