// scan_kernels.hip -- K2 (single/few-query cosine scan + per-wave top-k'), the
// chunk table of range-filtered scans, the select stage (merge + exact f64
// rescoring, optionally overlapped with the next scan: async select), and the
// cross-shard top-k merge.  K4 (threshold mode) lives in threshold.hip.
// gfx950 only (wave64, DPP, 16 B/lane row loads).
//
// Replaces the `for doc / for line: f32::cosine(q, e)` loop and the
// sort/take of search_documents (reference src/search/mod.rs:84-119) and the
// exact filtered scan behind Store::search_line_embeddings
// (src/workspace/store.rs:481-546).
//
// Data layout: corpus row-major f32 [rows x 256]; one row = 1024 B = one
// wave-wide 16 B/lane load (perfectly coalesced).  A wave owns U rows per
// iteration (U independent 1 KiB loads in flight), reduces ab = <q,c> and
// b2 = <c,c> with DPP butterflies, and keeps its best k' candidates in a
// lane-distributed sorted list (lane i = i-th best).  The f32 scan only
// NOMINATES candidates: the merge stage recomputes the distance of the final
// k' rows with f64 accumulation in index order (simsimd "accurate" formula),
// so returned distances/ordering do not depend on the reduction tree.
#include "common.h"
#include "device_utils.h"

namespace smt {

struct ScanParams {
    const float *corpus;
    const float *queries;  // [NQ x 256] (device)
    uint64_t n_virtual;    // unfiltered: rows to scan
    const uint64_t *chunk_table;  // FILTERED: one descriptor per chunk (row0 | valid rows << 32), see build_chunk_table_kernel
    uint64_t n_chunks;     // chunks to scan (FILTERED: table entries; else ceil(n_virtual / U))
    uint32_t kp;           // candidates kept per wave / per block (<= 64)
    uint32_t nq_active;    // queries of this pass that exist (<= NQ): a pass of 3 runs the 4-query kernel, slot 3 repeats query 0 and writes no list
    key_t64 *block_lists;  // [NQ][gridDim.x][kp]
    unsigned long long *stamps;  // optional (tuning key scan_debug_ptr): wall_clock64 per wave [start, loop end], per block [end]
    unsigned long long *flags;   // async select (or nullptr): [0] scan_done step, [1] select_done step, [2] blocks done, [3] timeout
    unsigned long long step;     // this launch's step number (>= 1)
    unsigned int *steal_ctr;     // STEAL: this launch's group counter (zero at launch), see scan_topk_kernel
    uint32_t steal_lr;           // STEAL: log2 of the rounds per group
    uint32_t steal_static;       // STEAL: a block's first steal_static groups are dealt statically (group g of block b = g * blocks + b)
};

// ---- async select: device-scope flags between the scan of step i (main stream) and its select (aux stream) ----
__device__ __forceinline__ unsigned long long flag_load(const unsigned long long *f)
{
    return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flag_store(unsigned long long *f, unsigned long long v)
{
    __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One thread waits until *f >= want.  Gives up after ~2 s of the 100 MHz wall clock and raises flags[3]
// (after which no kernel waits any more): a missing partner kernel must end in an error code at the next
// synchronise, never in a hung GPU.
__device__ __forceinline__ void flag_wait(unsigned long long *flags, int which, unsigned long long want)
{
    if (flag_load(flags + which) >= want) return;
    const unsigned long long t0 = wall_clock64();
    while (flag_load(flags + which) < want) {
        __builtin_amdgcn_s_sleep(16);
        if (flag_load(flags + 3) != 0ull) break;
        if (wall_clock64() - t0 > 200000000ull) { flag_store(flags + 3, 1ull); break; }
    }
}

// Range filter (path-subset search, src/workspace/store.rs:507-515): the rows to scan are the concatenation
// of sorted, disjoint row ranges.  Each range is cut into chunks of FILTER_CHUNK rows (the last one short),
// and ONE descriptor per chunk says where it starts and how many of its rows are valid, so the streaming
// kernels never search the range list: they read descriptor c (a wave-uniform 8-byte load, requested one
// pipeline stage before the rows) and stream row0 .. row0+cnt-1.  2 B per scanned row of extra traffic.
// (The first version binary-searched the ranges per ROW inside the scan loop: 3 dependent scalar loads in
// front of every row load made a 2-range search of 1 M rows 3x slower than the unfiltered one.)
__global__ void build_chunk_table_kernel(const smt_range *ranges, const uint64_t *chunk_prefix, uint32_t n_ranges,
                                         uint64_t n_chunks, uint64_t *table)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    uint32_t lo = 0, hi = n_ranges;  // invariant: chunk_prefix[lo] <= c < chunk_prefix[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (chunk_prefix[mid] <= c) lo = mid; else hi = mid;
    }
    const uint64_t row0 = ranges[lo].begin + (c - chunk_prefix[lo]) * FILTER_CHUNK;
    const uint64_t cnt = ranges[lo].end - row0 < (uint64_t)FILTER_CHUNK ? ranges[lo].end - row0 : (uint64_t)FILTER_CHUNK;
    table[c] = row0 | (cnt << 32);
}

int launch_build_chunk_table(smt_ctx *ctx, const smt_range *ranges, const uint64_t *chunk_prefix, uint32_t n_ranges,
                             uint64_t n_chunks, uint64_t *table)
{
    if (n_chunks == 0) return SMT_OK;
    hipLaunchKernelGGL(build_chunk_table_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, ctx->stream, ranges,
                       chunk_prefix, n_ranges, n_chunks, table);
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

// The same filter for the MFMA kernel (gemm_rowreg_kernel), whose unit is the ALIGNED 32-row tile -- the unit of the corpus' fp16
// operand image: one descriptor per tile that holds at least one wanted row, tile index | row mask << 32.  Ranges that meet
// inside a tile share its descriptor (tile_prefix counts a tile for the first range that touches it: search.cpp stage_ranges),
// so no tile is read twice however small the documents are.
__global__ void build_tile_table_kernel(const smt_range *ranges, const uint64_t *tile_prefix, uint32_t n_ranges, uint64_t n_vtiles,
                                        uint64_t *table)
{
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vtiles) return;
    uint32_t lo = 0, hi = n_ranges;  // first range i with tile_prefix[i + 1] > v (ranges that own no tile are skipped)
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tile_prefix[mid + 1] > v) hi = mid; else lo = mid + 1;
    }
    const uint32_t i = lo;
    const uint64_t first = ranges[i].begin >> 5;
    const bool shared = i > 0 && ((ranges[i - 1].end - 1) >> 5) == first;   // (empty ranges were dropped by the host)
    const uint64_t tile = first + (shared ? 1 : 0) + (v - tile_prefix[i]);
    const uint64_t t0 = tile << 5, t1 = t0 + 32;
    uint32_t mask = 0;
    for (uint32_t k = i; k < n_ranges && ranges[k].begin < t1; ++k) {
        const uint64_t b = ranges[k].begin > t0 ? ranges[k].begin : t0, e = ranges[k].end < t1 ? ranges[k].end : t1;
        if (e > b) mask |= (uint32_t)((~0ull >> (64 - (e - b))) << (b - t0));
    }
    table[v] = tile | ((uint64_t)mask << 32);
}

int launch_build_tile_table(smt_ctx *ctx, const smt_range *ranges, const uint64_t *tile_prefix, uint32_t n_ranges, uint64_t n_vtiles,
                            uint64_t *table)
{
    if (n_vtiles == 0) return SMT_OK;
    hipLaunchKernelGGL(build_tile_table_kernel, dim3((unsigned)((n_vtiles + 255) / 256)), dim3(256), 0, ctx->stream, ranges,
                       tile_prefix, n_ranges, n_vtiles, table);
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

// ------------------------------------------------------------------------- K2
// STEAL (round 6; unfiltered, 8-wave blocks): WHICH rows a block scans is decided while the kernel runs.  With the static deal every
// block owns the same number of rows, and the launch ends when the slowest block does: wall_clock64 stamps of 1 M-row launches
// (tools/scan_balance_async.py) show the blocks of one or two XCDs -- different ones from run to run -- finishing 10-14 us behind
// the median block (block ends 128 / 134 / 148 us: min / median / max), i.e. the kernel waits ~7 % of its time for a quarter of its
// blocks.  So the rounds (8 chunks = 32 rows, one per wave) are dealt in GROUPS of 2^steal_lr rounds, and only MOST of them up
// front: a block's first steal_static groups are static (group g of block b = g * gridDim.x + b, as before: nothing new in the
// loop but a compare), the last few per cent of the corpus are claimed group by group from a device counter -- STEAL_DEPTH groups
// ahead, by the wave that takes the first chunk of a group, the atomic's latency hidden behind that wave's row loads -- and
// published in an 8-entry LDS ring (group | base slot) that the waves' chunk claims look their rows up in.  Fast blocks simply
// come back for more.  (Everything dynamic was tried first: device-scope atomics on one address are served at ~65 per us on this
// part -- memory-side -- and 15 000 of them made the kernel 230 us instead of 151; a few hundred at the end are free.)
// RESULT, and why this is OFF by default (tuning key scan_steal = 0): the deal does what it should -- last block minus median
// block 2.7 us instead of 5-14 (profiles/r06_scan_balance.txt) -- and the launch is not a microsecond shorter (151.5 -> 152.5 us,
// profiles/r06_ab_steal.json): the median block moves UP to the last one, the last one does not come down.  The launch is bound by
// the part's AGGREGATE read rate (1.024 GB in 146-148 us of block time = 7.0 TB/s, the 0.88-0.90 of peak a 100 M-row launch shows
// too); blocks that finish early under the static deal are the ones that got a larger share of it, and their rows cost the same
// time whoever reads them.  Kept as an A/B switch with its parity test: it is the measurement that says the 1 M-row kernel has
// no imbalance left to recover, only its ramp.  Which block reduces a row cannot change
// the answer: the union of the blocks' k' best contains the k' best rows whatever the deal, and the select's threshold (the k'-th
// key of the union) is the k'-th smallest key of all rows either way.
constexpr int STEAL_DEPTH = 2;
template <int NQ, int U, bool NT, bool FILTERED, bool STEAL = false>
__global__ void __launch_bounds__(1024) scan_topk_kernel(ScanParams p)
{
    static_assert(!FILTERED || U == FILTER_CHUNK, "filtered scans use the chunk table's chunk size");
    static_assert(!STEAL || !FILTERED, "dynamic groups are for the unfiltered scan");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    key_t64 *s_keys = reinterpret_cast<key_t64 *>(smem_raw);  // [waves][64]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_block = blockDim.x >> 6;
    uint32_t *s_next = reinterpret_cast<uint32_t *>(s_keys + waves_per_block * 64);  // the block's next unclaimed chunk
    const uint64_t wave_global = (uint64_t)blockIdx.x * waves_per_block + wave;
    const int kp = (int)p.kp;
    if (p.stamps && lane == 0) p.stamps[wave_global * 2] = wall_clock64();

    f32x4 q[NQ];
    float rq[NQ];
    bool qz[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        q[n] = reinterpret_cast<const f32x4 *>(p.queries + ((uint32_t)n < p.nq_active ? n : 0) * 256)[lane];
        const float a2 = wave_sum(q[n].x * q[n].x + q[n].y * q[n].y + q[n].z * q[n].z + q[n].w * q[n].w);
        qz[n] = (a2 == 0.0f);
        rq[n] = qz[n] ? 0.0f : __frsqrt_rn(a2);
    }

    // lane-distributed sorted candidate lists
    float ld[NQ];
    uint32_t lr[NQ];
    float thr_d[NQ];
    uint32_t thr_r[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        ld[n] = __builtin_inff();
        lr[n] = 0xFFFFFFFFu;
        thr_d[n] = __builtin_inff();
        thr_r[n] = 0xFFFFFFFFu;
    }

    // The block owns the chunks (U rows each) c(t) = ((t / waves) * gridDim.x + blockIdx.x) * waves + t % waves,
    // t = 0, 1, ... -- the same rows a static grid-stride deal would give it -- but its waves CLAIM them from
    // an LDS counter.  The CU's arbiter favours the older wave of each SIMD: with a static deal waves 0-3
    // finished 12 us before waves 4-7 (of 145) and the CU ran half empty at the end.  Which wave reduces a
    // chunk cannot change the block's top-k' (same row set), so the output is still a function of the data.
    const uint64_t n_chunks = p.n_chunks;
    // the table was written by an earlier kernel and is read-only here: constant address space => scalar loads
    // (a plain global load is a VECTOR load that returns in order with the row loads and drains the pipeline)
    const const_u64_ptr const_table = (const_u64_ptr)(uintptr_t)p.chunk_table;
    auto chunk_id = [&](uint32_t t) -> uint64_t {
        const uint64_t c = ((uint64_t)(t / waves_per_block) * gridDim.x + blockIdx.x) * waves_per_block + t % waves_per_block;
        return uniform_u64(c);  // the division runs on the VALU: tell the compiler the result is wave-uniform (scalar loads)
    };
    auto claim = [&]() -> uint32_t {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(s_next, 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    };
    // descriptor of chunk c: first corpus row | valid rows << 32 (0 rows beyond the end)
    auto fetch_desc = [&](uint64_t c) -> uint64_t {
        return c < n_chunks ? const_table[c] : 0ull;  // wave-uniform address in the constant address space: s_load
    };
    auto issue_loads = [&](uint64_t desc, f32x4 (&c)[U], uint32_t (&row)[U]) {
        const uint32_t row0 = (uint32_t)desc, cnt = (uint32_t)(desc >> 32);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            row[j] = row0 + ((uint32_t)j < cnt ? (uint32_t)j : 0u);  // rows beyond cnt re-read row0 (result discarded)
            const f32x4 *src = reinterpret_cast<const f32x4 *>(p.corpus + (uint64_t)row[j] * 256) + lane;
            c[j] = NT ? __builtin_nontemporal_load(src) : *src;
        }
    };
    // STEAL: ring[g & 7] = (group g of this block | its base slot / 2^steal_lr); the first STEAL_DEPTH groups are static
    uint2 *s_ring = reinterpret_cast<uint2 *>(s_next + 2);
    if (threadIdx.x == 0) *s_next = (uint32_t)waves_per_block;  // chunks 0..waves-1 are dealt: wave w starts on chunk w
    if constexpr (STEAL) {
        if (threadIdx.x < 8) s_ring[threadIdx.x] = make_uint2(0xFFFFFFFFu, 0u);
    }
    __syncthreads();

    // One row against the NQ queries.  `valid` is wave-uniform and gates only the (rare) insert, never the
    // arithmetic: the four rows' DPP chains must stay free to interleave.  How `valid` is WRITTEN matters: with
    // "j < rows_in_chunk" clang hoists a branch in front of every row's reduction (+2 us per 1 M-row launch,
    // A/B on one box); "(v0 + j) < n" below does not trigger that, so the unfiltered loop keeps that form.
#define SMT_REDUCE_ROW(cj, rj, valid_expr)                                                                        \
    do {                                                                                                          \
        const bool valid = (valid_expr);                                                                          \
        const float b2 = wave_sum((cj).x * (cj).x + (cj).y * (cj).y + (cj).z * (cj).z + (cj).w * (cj).w);         \
        _Pragma("unroll") for (int n = 0; n < NQ; ++n) {                                                          \
            const float ab = wave_sum((cj).x * q[n].x + (cj).y * q[n].y + (cj).z * q[n].z + (cj).w * q[n].w);     \
            const float d = dist_f32(ab, b2, rq[n], qz[n]);                                                       \
            const uint32_t r = (rj);                                                                              \
            if (valid && (d < thr_d[n] || (d == thr_d[n] && r < thr_r[n]))) {                                     \
                /* insert (d, r) keeping (distance asc, row asc) order */                                         \
                const bool less = (ld[n] < d) || (ld[n] == d && lr[n] < r);                                       \
                const int pos = __popcll(__ballot(less));                                                         \
                const float sd = dpp_f<DPP_WAVE_SHR1>(ld[n]);                                                     \
                const uint32_t sr = dpp_u<DPP_WAVE_SHR1>(lr[n]);                                                  \
                if (lane > pos) { ld[n] = sd; lr[n] = sr; }                                                       \
                else if (lane == pos) { ld[n] = d; lr[n] = r; }                                                   \
                thr_d[n] = readlane_f(ld[n], kp - 1);                                                             \
                thr_r[n] = (uint32_t)__builtin_amdgcn_readlane((int)lr[n], kp - 1);                               \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)

    // Four rows per chunk (the default U): the four rows' partial products are reduced TOGETHER -- wave_sum4 leaves the sum
    // of row j in the lanes with lane % 4 == j -- once for <c, c> and once per query, the distance is computed lane-parallel (four
    // (row, query) pairs per instruction) and one ballot finds the pairs under the query's threshold; only those (rare once the
    // lists have filled) are read out and inserted.  Per chunk and query 15 + ~14 instructions instead of 4 x (11 + ~14): with the
    // row-at-a-time form the reductions made two queries cost 1.4 x and four 2.3 x one query's pass (instruction issue, not HBM).
    // VALID(j): wave-uniform, is row j of the chunk a real row.
#define SMT_REDUCE_CHUNK4(cq, rq4, VALID)                                                                         \
    do {                                                                                                          \
        const int jj = lane & 3;                                                                                  \
        const bool ok_0 = VALID(0), ok_1 = VALID(1), ok_2 = VALID(2), ok_3 = VALID(3);                            \
        const bool valid_mine = jj == 0 ? ok_0 : jj == 1 ? ok_1 : jj == 2 ? ok_2 : ok_3;                          \
        const uint32_t r_mine = jj == 0 ? (rq4)[0] : jj == 1 ? (rq4)[1] : jj == 2 ? (rq4)[2] : (rq4)[3];          \
        float pb[4];                                                                                              \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                             \
            pb[j] = (cq)[j].x * (cq)[j].x + (cq)[j].y * (cq)[j].y + (cq)[j].z * (cq)[j].z + (cq)[j].w * (cq)[j].w; \
        const float b2 = wave_sum4(pb[0], pb[1], pb[2], pb[3], lane);                                             \
        _Pragma("unroll") for (int n = 0; n < NQ; ++n) {                                                          \
            float pa[4];                                                                                          \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                         \
                pa[j] = (cq)[j].x * q[n].x + (cq)[j].y * q[n].y + (cq)[j].z * q[n].z + (cq)[j].w * q[n].w;        \
            const float ab = wave_sum4(pa[0], pa[1], pa[2], pa[3], lane);                                         \
            const float d4 = dist_f32(ab, b2, rq[n], qz[n]);                                                      \
            const bool cand = valid_mine && (d4 < thr_d[n] || (d4 == thr_d[n] && r_mine < thr_r[n]));             \
            uint32_t pending = (uint32_t)__ballot(cand) & 0xFu;                                                   \
            while (pending) {                                                                                     \
                const int j = __builtin_ctz(pending);                                                             \
                pending &= pending - 1;                                                                           \
                const float d = readlane_f(d4, j);                                                                \
                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)r_mine, j);                           \
                if (d < thr_d[n] || (d == thr_d[n] && r < thr_r[n])) {   /* (an earlier insert may have moved the threshold) */ \
                    const bool less = (ld[n] < d) || (ld[n] == d && lr[n] < r);                                   \
                    const int pos = __popcll(__ballot(less));                                                     \
                    const float sd = dpp_f<DPP_WAVE_SHR1>(ld[n]);                                                 \
                    const uint32_t sr = dpp_u<DPP_WAVE_SHR1>(lr[n]);                                              \
                    if (lane > pos) { ld[n] = sd; lr[n] = sr; }                                                   \
                    else if (lane == pos) { ld[n] = d; lr[n] = r; }                                               \
                    thr_d[n] = readlane_f(ld[n], kp - 1);                                                         \
                    thr_r[n] = (uint32_t)__builtin_amdgcn_readlane((int)lr[n], kp - 1);                           \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
    // (one query too: 2 x 15 instead of 8 x 11 reduction instructions per chunk -- A/B of two builds on one box, tools/ab_libs.sh:
    // 155.6 -> 152.8 us per 1 M-row launch, 0.822 -> 0.838 of HBM)
    constexpr bool CHUNK4 = U == 4;

    // Software pipeline: the rows of the next chunk are requested right after the current chunk's rows arrived
    // (the register copy below waits for them) and fly during the current reduction; the chunk after that is
    // claimed meanwhile.  Measured alternatives, all slower at 1 M rows: a ping-pong over two register buffers
    // (two chunks in flight per wave: 160 vs 149 us), 16 waves/CU, U = 8 -- more than ~32 KiB of requests in
    // flight per CU costs bandwidth.
    if constexpr (!FILTERED) {
        auto issue_rows = [&](uint64_t v0, f32x4 (&c)[U], uint32_t (&row)[U]) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                uint64_t v = v0 + j;
                if (v >= p.n_virtual) v = p.n_virtual - 1;  // clamp (result discarded)
                row[j] = (uint32_t)v;
                const f32x4 *src = reinterpret_cast<const f32x4 *>(p.corpus + (uint64_t)row[j] * 256) + lane;
                c[j] = NT ? __builtin_nontemporal_load(src) : *src;
            }
        };
        // STEAL: a group this wave has asked the device counter for (lane 0 holds the answer once the atomic has returned)
        uint32_t pend_group = 0xFFFFFFFFu, pend_val = 0;
        auto chunk_v0 = [&](uint32_t t) -> uint64_t {
            if constexpr (STEAL) {
                // (8-wave blocks) round r = t / 8 of this block lies in its group g = r >> lr; the group's base comes from the ring
                const uint32_t r = t >> 3, g = r >> p.steal_lr, rho = r & ((1u << p.steal_lr) - 1u);
                uint32_t base;
                if (g < p.steal_static) {
                    base = g * gridDim.x + blockIdx.x;   // the static part of the deal: no look-up, no atomic
                } else {
                    // (a wave never waits while it holds an answer others may be waiting for: twice in a row only in the prologue)
                    if (pend_group != 0xFFFFFFFFu) {
                        if (lane == 0) s_ring[pend_group & 7u] = make_uint2(pend_group, p.steal_static * gridDim.x + pend_val);
                        pend_group = 0xFFFFFFFFu;
                    }
                    uint32_t tag, spins = 0;
                    do {
                        // (one address for the wave: a broadcast 8-byte read, so that tag and base come from ONE publish)
                        const unsigned long long e = *reinterpret_cast<volatile unsigned long long *>(s_ring + (g & 7u));
                        tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)e);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(e >> 32));
                        if (++spins == (1u << 26)) __builtin_trap();   // (seconds: a publish that never comes must end in an error, not in a hung GPU)
                    } while (tag != g);   // (published STEAL_DEPTH groups ahead: the wait is for a wave that is late with its publish)
                }
                const uint64_t slot = ((uint64_t)base << p.steal_lr) + rho;
                const uint64_t v = (slot * 8u + (t & 7u)) * U;
                // the wave that takes the FIRST chunk of group g fetches group g + STEAL_DEPTH if that one is dynamic (unless its own
                // group lies past the end: then so do all later ones, and the ring gets a base that says so without asking the counter)
                if ((t & ((8u << p.steal_lr) - 1u)) == 0u && g + STEAL_DEPTH >= p.steal_static) {
                    if (pend_group != 0xFFFFFFFFu) {
                        if (lane == 0) s_ring[pend_group & 7u] = make_uint2(pend_group, p.steal_static * gridDim.x + pend_val);
                        pend_group = 0xFFFFFFFFu;
                    }
                    if (v < p.n_virtual) {
                        if (lane == 0) pend_val = atomicAdd(p.steal_ctr, 1u);
                        pend_group = g + STEAL_DEPTH;
                    } else if (lane == 0) {
                        s_ring[(g + STEAL_DEPTH) & 7u] = make_uint2(g + STEAL_DEPTH, 0x0FFFFFFFu);
                    }
                }
                return uniform_u64(v);
            } else {
                return (((uint64_t)(t / waves_per_block) * gridDim.x + blockIdx.x) * waves_per_block + t % waves_per_block) * U;
            }
        };
        // ... and publishes it once the answer is there: called where the wave has just waited for its row loads, which the atomic
        // was issued in front of
        auto publish = [&]() {
            if constexpr (STEAL) {
                if (pend_group != 0xFFFFFFFFu) {
                    if (lane == 0) s_ring[pend_group & 7u] = make_uint2(pend_group, p.steal_static * gridDim.x + pend_val);
                    pend_group = 0xFFFFFFFFu;
                }
            }
        };
        f32x4 cn[U];
        uint32_t rown[U];
        uint64_t v0 = chunk_v0((uint32_t)wave);
        uint64_t v0n = 0;
        if (v0 < p.n_virtual) {
            issue_rows(v0, cn, rown);
            v0n = chunk_v0(claim());
        }
        while (v0 < p.n_virtual) {
            f32x4 c[U];
            uint32_t row[U];
#pragma unroll
            for (int j = 0; j < U; ++j) { c[j] = cn[j]; row[j] = rown[j]; }
            publish();
            if (v0n < p.n_virtual) issue_rows(v0n, cn, rown);
            const uint64_t v0nn = chunk_v0(claim());
            if constexpr (CHUNK4) {
#define SMT_VALID_UNF(j) ((v0 + (j)) < p.n_virtual)
                SMT_REDUCE_CHUNK4(c, row, SMT_VALID_UNF);
#undef SMT_VALID_UNF
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) SMT_REDUCE_ROW(c[j], row[j], (v0 + j) < p.n_virtual);
            }
            v0 = v0n;
            v0n = v0nn;
        }
    } else {
        // range-filtered: one more stage in front -- [claim chunk c+2, request its descriptor (scalar load)] ->
        // [request the rows of c+1] -> [reduce c]
        f32x4 cn[U];
        uint32_t rown[U];
        uint64_t cA = chunk_id((uint32_t)wave), cB = n_chunks, cC = n_chunks;
        uint64_t dA = fetch_desc(cA), dB = 0;
        if (cA < n_chunks) {
            issue_loads(dA, cn, rown);
            cB = chunk_id(claim());
            dB = fetch_desc(cB);
            cC = chunk_id(claim());
        }
        while (cA < n_chunks) {
            f32x4 c[U];
            uint32_t row[U];
#pragma unroll
            for (int j = 0; j < U; ++j) { c[j] = cn[j]; row[j] = rown[j]; }
            if (cB < n_chunks) issue_loads(dB, cn, rown);
            const uint64_t dC = fetch_desc(cC);
            const uint64_t cD = chunk_id(claim());
            const uint64_t end = (dA & 0xFFFFFFFFull) + (dA >> 32);  // first row past this chunk
            if constexpr (CHUNK4) {
#define SMT_VALID_FIL(j) ((dA & 0xFFFFFFFFull) + (j) < end)
                SMT_REDUCE_CHUNK4(c, row, SMT_VALID_FIL);
#undef SMT_VALID_FIL
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) SMT_REDUCE_ROW(c[j], row[j], (dA & 0xFFFFFFFFull) + j < end);
            }
            cA = cB; dA = dB;
            cB = cC; dB = dC;
            cC = cD;
        }
    }
#undef SMT_REDUCE_ROW
#undef SMT_REDUCE_CHUNK4

    if (p.stamps && lane == 0) p.stamps[wave_global * 2 + 1] = wall_clock64();

    // async select: this launch fills the list buffer that the select of step-2 read, and the aux stream must
    // not fall behind: wait for the select of step-1 (it started when this scan did and takes ~14 us -- by the
    // time a block gets here the flag is long up, the wait normally reads it once)
    if (p.flags && threadIdx.x == 0 && p.step >= 2) flag_wait(p.flags, 1, p.step - 1);

    // block merge: rank every wave's candidates among all of the block's.  ONE barrier for all the queries of the pass, and no global
    // store in front of it: every query has its own LDS region (the loop's chunk counter and ring are dead by now), and every slot of
    // a block list is written exactly once -- its key, or the padding from the first slot no key ranks into.  (Until round 6 each
    // query took two barriers with the padding stored between them; a __syncthreads() waits for the wave's outstanding global
    // stores, so every query -- the one-query headline launch included -- paid a store round trip or two in its tail: a one-row
    // search cost 4.4 / 9.4 / 15 / 18.5 us of scan kernel with 1 / 2 / 3 / 4 queries.)
    __syncthreads();
    unsigned int *s_valid = reinterpret_cast<unsigned int *>(s_keys + (size_t)NQ * waves_per_block * 64);   // [NQ][waves] valid keys per wave list
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const bool have = lane < kp && lr[n] != 0xFFFFFFFFu;
        s_keys[((size_t)n * waves_per_block + wave) * 64 + lane] = have ? make_key(ld[n], lr[n]) : KEY_PAD;
        const unsigned int cnt = (unsigned int)__popcll(__ballot(have));
        if (lane == 0) s_valid[n * waves_per_block + wave] = cnt;
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        if ((uint32_t)n >= p.nq_active) break;   // (uniform over the block)
        const key_t64 *keys_n = s_keys + (size_t)n * waves_per_block * 64;
        key_t64 *out = p.block_lists + ((size_t)n * gridDim.x + blockIdx.x) * kp;
        const key_t64 mine = keys_n[wave * 64 + lane];
        if (mine != KEY_PAD) {
            int rank = 0;
            for (int w = 0; w < waves_per_block; ++w)
                for (int i = 0; i < kp; ++i) rank += (keys_n[w * 64 + i] < mine) ? 1 : 0;
            if (rank < kp) out[rank] = mine;
        }
        if (wave == 0 && lane < kp) {   // the padding: slots [valid keys of the block, kp)
            unsigned int total = 0;
            for (int w = 0; w < waves_per_block; ++w) total += s_valid[n * waves_per_block + w];
            if ((unsigned int)lane >= total) out[lane] = KEY_PAD;
        }
    }
    if (p.flags) {
        // publish: the block's list stores are ordered before its barrier (workgroup scope); ONE thread then
        // does the agent-scope release (one L2 write-back per block -- a __threadfence() in every thread cost
        // 50 us per launch), and the LAST block raises scan_done
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long prev = __hip_atomic_fetch_add(p.flags + 2, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == (unsigned long long)gridDim.x - 1) {
                flag_store(p.flags + 2, 0ull);
                __hip_atomic_store(p.flags + 0, p.step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (p.stamps && threadIdx.x == 0) p.stamps[(uint64_t)gridDim.x * waves_per_block * 2 + blockIdx.x] = wall_clock64();
}

// --------------------------------------------------- exact f64 distance (A5)
// Bit-for-bit the oracle's orc_cosine_f32_accurate: f64 accumulators, index
// order, then cos_finish.  The product of two f32 values is exact in f64, so
// fma(a, b, acc) rounds exactly like acc + a*b: letting the compiler contract
// to v_fma_f64 cannot change a bit (the oracle is built with contraction off).
// cos_finish of the oracle.  Contraction OFF here: 1 - (ab*ra)*rb must round the
// products before the subtraction exactly as the CPU code does.
__device__ __forceinline__ double cos_finish_exact(double ab, double a2, double b2)
{
#pragma clang fp contract(off)
    if (a2 == 0.0 && b2 == 0.0) return 0.0;
    if (ab == 0.0) return 1.0;
    const double ra = 1.0 / sqrt(a2);
    const double rb = 1.0 / sqrt(b2);
    const double t = ab * ra;
    const double u = t * rb;
    const double unclipped = 1.0 - u;
    return unclipped > 0.0 ? unclipped : 0.0;
}

// ab and b2 chains for one row (index order), a2 supplied by the caller (same for every row).
template <typename QP, typename RP>
__device__ __forceinline__ void exact_sums(QP q4, RP r4, double &ab_out, double &b2_out)
{
    double ab = 0.0, b2 = 0.0;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) {
        const f32x4 a = q4[i];
        const f32x4 b = r4[i];
        const double ax = a.x, ay = a.y, az = a.z, aw = a.w;
        const double bx = b.x, by = b.y, bz = b.z, bw = b.w;
        ab = ab + ax * bx; b2 = b2 + bx * bx;
        ab = ab + ay * by; b2 = b2 + by * by;
        ab = ab + az * bz; b2 = b2 + bz * bz;
        ab = ab + aw * bw; b2 = b2 + bw * bw;
    }
    ab_out = ab;
    b2_out = b2;
}

template <typename QP>
__device__ __forceinline__ double exact_norm2(QP q4)
{
    double a2 = 0.0;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) {
        const f32x4 a = q4[i];
        const double ax = a.x, ay = a.y, az = a.z, aw = a.w;
        a2 = a2 + ax * ax; a2 = a2 + ay * ay; a2 = a2 + az * az; a2 = a2 + aw * aw;
    }
    return a2;
}

template <typename QP, typename RP>
__device__ __forceinline__ double exact_distance(QP q4, RP r4)
{
    double ab, b2;
    exact_sums(q4, r4, ab, b2);
    return cos_finish_exact(ab, exact_norm2(q4), b2);
}

__global__ void rescore_rows_kernel(const float *corpus, const float *query, const uint32_t *rows,
                                    uint64_t n, double *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = exact_distance(reinterpret_cast<const f32x4 *>(query),
                                reinterpret_cast<const f32x4 *>(corpus + (uint64_t)rows[i] * 256));
}

// the same for the rows of MANY queries in one launch: row i belongs to query qidx[i] of the [n_queries][256] block
__global__ void rescore_rows_multi_kernel(const float *corpus, const float *queries, const uint32_t *rows, const uint32_t *qidx,
                                          uint64_t n, double *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = exact_distance(reinterpret_cast<const f32x4 *>(queries + (uint64_t)qidx[i] * 256),
                                reinterpret_cast<const f32x4 *>(corpus + (uint64_t)rows[i] * 256));
}

// dst[i] = src[idx[i]] for rows of 256 f32: one wave per row
__global__ void gather_rows256_kernel(const float *src, const uint32_t *idx, uint32_t n, float *dst)
{
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i < n)
        reinterpret_cast<f32x4 *>(dst + (size_t)i * 256)[threadIdx.x & 63] =
            reinterpret_cast<const f32x4 *>(src + (size_t)idx[i] * 256)[threadIdx.x & 63];
}

// ------------------------------------------------ select: prune, rank, rescore
// Input: n_lists sorted lists (one per scan block) of kp keys each, ascending,
// padded with KEY_PAD; valid keys are distinct (distinct rows).  We need the kp
// smallest keys of the union WITHOUT sorting M = n_lists*kp keys.
//
// Pruning bound.  For a column j (1-based) let c_j = ceil(kp / j) and let
// tau_j be the c_j-th smallest of the lists' j-th entries.  At least c_j lists
// hold >= j keys <= tau_j, i.e. >= kp keys of the union are <= tau_j, so the
// union's kp-th smallest key is <= tau_j.  tau = min over a dyadic set of
// columns {1,2,4,..,kp}.  Keys > tau can be dropped.  Survivors are few: if
// m_b = #keys of list b that are <= tau then #{b : m_b >= j} <= c_j' for the
// largest used column j' <= j, hence S = sum_b m_b <= sum_j' (gap_j' * c_j')
// which is <= kp * (log2(kp) + 2) in the WORST case (<= 512 for kp <= 72), and
// about kp + 2 on uncorrelated data (column 1 alone is then nearly tight).
// Survivors are ranked by counting (S^2 / threads compares), no sort.
constexpr int SEL_THREADS = 1024;
constexpr int SEL_MAX_LISTS = 512;
constexpr int SEL_MAX_COLS = 8;
constexpr int SEL_SURV_CAP = 1024;
constexpr int SEL_FEW_KEYS = 512;    // up to this many keys over all lists the select ranks them all (no column bounds)
constexpr int ROW_STRIDE_F4 = 65;  // LDS row stride in float4 (1040 B): conflict-free b128 reads
constexpr int PROD_STRIDE = 258;   // LDS row stride of the f64 product tables (2064 B)
constexpr int SEL_FAST_KP = 34;    // product tables fit the 160 KiB LDS up to this k' (k <= 26)

// c-th smallest key of `col[0..L)` (PAD = missing), published with atomicMin into *tau.
// Executed by ONE wave; slots beyond L hold PAD which never satisfies the predicates below.
template <int NS>
__device__ __forceinline__ void column_tau(const key_t64 *col, int L, int c, key_t64 *tau)
{
    const int lane = threadIdx.x & 63;
    uint32_t vh[NS], vl[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int b = lane + u * 64;
        const key_t64 x = b < L ? col[b] : KEY_PAD;
        vh[u] = (uint32_t)(x >> 32);
        vl[u] = (uint32_t)x;
    }
    uint32_t lo = 0, hi = 0xFFFFFFFFu;
    while (lo < hi) {  // smallest distance t with #{vh <= t} >= c   (mid <= 0xFFFFFFFE: PAD never counts)
        const uint32_t mid = lo + ((hi - lo) >> 1);
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < NS; ++u) cnt += __popcll(__ballot(vh[u] <= mid));
        if (cnt >= c) hi = mid; else lo = mid + 1;
    }
    if (lo == 0xFFFFFFFFu) return;  // fewer than c entries: no bound from this column
    const uint32_t dstar = lo;
    int below = 0, ties = 0;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        below += __popcll(__ballot(vh[u] < dstar));
        ties += __popcll(__ballot(vh[u] == dstar));
    }
    const int need = c - below;  // need-th smallest row among the entries at distance dstar (>= 1)
    uint32_t rlo = 0xFFFFFFFFu;  // every tie needed: tau = (dstar, max row)
    if (ties != need) {
        rlo = 0;
        uint32_t rhi = 0xFFFFFFFFu;
        while (rlo < rhi) {
            const uint32_t mid = rlo + ((rhi - rlo) >> 1);
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < NS; ++u) cnt += __popcll(__ballot(vh[u] == dstar && vl[u] <= mid));
            if (cnt >= need) rhi = mid; else rlo = mid + 1;
        }
    }
    if (lane == 0) atomicMin(tau, ((key_t64)dstar << 32) | (key_t64)rlo);
}

struct FinalParams {
    const float *corpus;
    const float *queries;
    const key_t64 *lists;  // query qi's lists start at lists + qi*list_stride: [n_lists][kp]
    uint64_t list_stride;  // in keys
    uint32_t n_lists;      // <= SEL_MAX_LISTS
    uint32_t kp;
    uint32_t k_out;
    int ws_threshold;
    float ws_thr_score;
    uint64_t row_base;
    uint64_t *out_rows;   // [nq][k_out]
    double *out_dist;     // [nq][k_out]
    uint64_t *out_counts; // [nq] or nullptr
    uint64_t out_stride;  // words between consecutive queries' output lists (>= k_out)
    double f32_err;       // > 0: exactness certificate (SelectArgs::f32_err)
    uint64_t *out_uncertain;       // [nq] or nullptr
    unsigned int *out_status;      // [nq] or nullptr (SelectArgs::out_status)
    unsigned long long *status;    // the context's sticky "uncertain selects" counter
    unsigned long long *dbg;  // optional: s_memtime stamps of query 0's phases (tuning key select_debug_ptr)
    unsigned long long *flags;  // async select (or nullptr), see ScanParams
    unsigned long long step;
    const unsigned int *overflow;  // SelectArgs::overflow
    // delivery by the last block to finish (common.h Delivery; host_flag == nullptr: none)
    const unsigned long long *dev_out;
    unsigned long long *host_out;
    unsigned long long *host_flag;
    unsigned long long *done;
    unsigned long long seq;
    uint32_t out_words;
};

#define SEL_STAMP(i) do { if (p.dbg && blockIdx.x == 0 && threadIdx.x == 0) p.dbg[i] = __builtin_readcyclecounter(); } while (0)

// One block per query.
// KREG = keys a thread holds in registers: 8 covers n_lists * k' <= 8192 (every k <= 24 at 256 lists) in 70
// VGPRs, so that the 16-wave block fits on a CU NEXT TO a scan block (async select); 36 covers the maximum
// (512 lists x 72) and takes the whole register file.
// OVF: the batched path's instantiation, which also reads the candidate-buffer overflow flag (SelectArgs::overflow).  A template
// parameter and not a null test: with the flag code in it final_select_kernel<8> allocated 106 VGPRs instead of 90 and no longer
// fitted on a CU NEXT TO a scan block (4 waves/SIMD x 96 + the scan's 2 x 56 <= 512) -- the async select of the one-query pipeline
// then waited for scan blocks to leave and a 1 M-row step went from 153 to 204 us (found by the round-5 closing bench run).
template <int KREG, bool OVF>
__device__ __forceinline__ void final_select_body(const FinalParams &p, const uint32_t qi, unsigned char *smem_raw)
{
    const int kp = (int)p.kp;
    const int L = (int)p.n_lists;
    // LDS carve (all offsets multiples of 16; no static LDS in this kernel): small arrays first, then one big
    // region used twice -- [candidate-row tiles | survivors | column entries] until the best k' are known,
    // then (fast path) the exact-product tables of the rescoring stage.
    unsigned int *s_srank = reinterpret_cast<unsigned int *>(smem_raw);        // [96]
    unsigned int *s_rank = s_srank + 96;                                        // [kp+pad] final ranks
    double *s_qd = reinterpret_cast<double *>(s_rank + 80);                     // [256] query as f64 (slow path)
    key_t64 *s_best = reinterpret_cast<key_t64 *>(s_qd + 256);                  // [kp] (+pad to even)
    double *s_d = reinterpret_cast<double *>(s_best + ((kp + 1) & ~1));        // [kp]
    double *s_b2 = s_d + ((kp + 1) & ~1);                                       // [kp] row norms^2 (fast path)
    uint32_t *s_r = reinterpret_cast<uint32_t *>(s_b2 + ((kp + 1) & ~1));      // [kp]
    key_t64 *s_tau = reinterpret_cast<key_t64 *>(s_r + ((kp + 3) & ~3));       // [1]
    double *s_a2 = reinterpret_cast<double *>(s_tau + 1);                       // [1] query norm^2
    unsigned int *s_cnt = reinterpret_cast<unsigned int *>(s_a2 + 1);          // [0]=survivors [1]=valid [2]=max |q_i| bits [3]=this block delivers
    unsigned char *s_big = reinterpret_cast<unsigned char *>(s_cnt + 4);
    f32x4 *s_rows = reinterpret_cast<f32x4 *>(s_big);                           // [kp+1][65] float4
    key_t64 *s_surv = reinterpret_cast<key_t64 *>(s_rows + (size_t)(kp + 1) * ROW_STRIDE_F4);  // [SEL_SURV_CAP]
    key_t64 *s_col = s_surv + SEL_SURV_CAP;                                      // [ncols][L] column entries
    double *s_P = reinterpret_cast<double *>(s_big);                            // [kp][PROD_STRIDE] q_i * c_i (exact in f64)
    double *s_B = s_P + (size_t)kp * PROD_STRIDE;                               // [kp][PROD_STRIDE] c_i * c_i
    double *s_Q = s_B + (size_t)kp * PROD_STRIDE;                               // [256]            q_i * q_i
    const bool fast_rescore = kp <= SEL_FAST_KP;

    const key_t64 *lists = p.lists + (size_t)qi * p.list_stride;
    if (p.flags) {
        // launched on the aux stream while the scan of this step may still be running on the main stream
        if (threadIdx.x == 0) {
            flag_wait(p.flags, 0, p.step);
            // acquire at agent scope by ONE thread (invalidates this CU's vector L1 and the XCD's non-coherent
            // L2 lines: the lists were written through other XCDs' L2s); the barrier hands it to the block
            (void)__hip_atomic_load(p.flags + 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    SEL_STAMP(0);

    static_assert(KREG == 8 || KREG == (SEL_MAX_LISTS * 72 + SEL_THREADS - 1) / SEL_THREADS, "8 or 36");
    if (L == 1) {
        // ONE list per query (the batched kernel's level select left the k' best of every query sorted at the head of its buffer;
        // an index search with one probed segment): it IS the answer's candidate set -- no column bounds, no compaction, no ranks.
        // (Round 5: those three phases were ~7 us of dependent latencies in front of every batched call's rescoring, 10 % of a
        // small batch over a small corpus.)
        if (threadIdx.x == 0) { *s_tau = KEY_PAD; s_cnt[0] = 0; s_cnt[1] = 0; s_cnt[2] = 0; }
        for (int t = threadIdx.x; t < kp; t += blockDim.x) { s_best[t] = lists[t]; s_rank[t] = 0; }
        __syncthreads();
    } else {
    // ---- ONE global-latency phase: every key of the L lists goes to registers (<= 36 per thread)
    const int M = L * kp;
    const int n_round = (M + SEL_THREADS - 1) / SEL_THREADS;  // block-uniform
    key_t64 kreg[KREG];
    if (KREG == 8 || n_round <= 8) {  // common case (k <= 24): 8 back-to-back loads, one wait
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = (int)threadIdx.x + i * SEL_THREADS;
            const key_t64 x = lists[e < M ? e : M - 1];  // clamped: no branch between the loads
            kreg[i] = e < M ? x : KEY_PAD;
        }
#pragma unroll
        for (int i = 8; i < KREG; ++i) kreg[i] = KEY_PAD;
    } else {
#pragma unroll
        for (int i = 0; i < KREG; ++i) {
            const int e = (int)threadIdx.x + i * SEL_THREADS;
            const key_t64 x = lists[e < M ? e : M - 1];
            kreg[i] = e < M ? x : KEY_PAD;
        }
    }

    // dyadic columns 1,2,4,.. < kp, plus kp: column jj holds entry (1 << jj) except the last = kp
    int ncols = 1;
    for (int j = 1; j < kp && ncols < SEL_MAX_COLS; j <<= 1) ++ncols;
    auto col_of = [&](int jj) { return jj < ncols - 1 ? (1 << jj) : kp; };
    // the column entries are fetched in the same latency window as the keys (<= 4 per thread)
    // FEW keys in all (a small corpus: 16 blocks x 11 keys for 1000 rows): no pruning bound -- every key is a survivor and the ranks
    // below order them directly.  (The column bounds are ~3 us of dependent bisection steps, a quarter of this stage, to prune a set
    // that is already small.)
    const bool few = M <= SEL_FEW_KEYS;   // block-uniform
    key_t64 cval[(SEL_MAX_COLS * SEL_MAX_LISTS) / SEL_THREADS];
    const int n_col_entries = few ? 0 : ncols * L;
    if (!few) {
#pragma unroll
        for (int u = 0; u < (SEL_MAX_COLS * SEL_MAX_LISTS) / SEL_THREADS; ++u) {
            int t = (int)threadIdx.x + u * SEL_THREADS;
            const bool ok = t < n_col_entries;
            t = ok ? t : 0;
            const int jj = t / L, bb = t - jj * L;
            const key_t64 x = lists[(size_t)bb * kp + (col_of(jj) - 1)];
            cval[u] = ok ? x : KEY_PAD;
        }
    }
    if (threadIdx.x == 0) { *s_tau = KEY_PAD; s_cnt[0] = 0; s_cnt[1] = 0; s_cnt[2] = 0; }
    for (int t = threadIdx.x; t < kp; t += blockDim.x) { s_best[t] = KEY_PAD; s_rank[t] = 0; }
    if (!few) {
#pragma unroll
        for (int u = 0; u < (SEL_MAX_COLS * SEL_MAX_LISTS) / SEL_THREADS; ++u) {
            const int t = (int)threadIdx.x + u * SEL_THREADS;
            if (t < n_col_entries) s_col[t] = cval[u];
        }
    }
    __syncthreads();
    SEL_STAMP(1);

    // tau_j = c_j-th smallest key of column j.  One wave per column: the column sits in registers
    // (4 or 8 keys per lane); bisection on the 32 distance bits with ballot counts, then -- only if
    // several entries share that distance -- on the row bits.  No sort, no O(L^2) ranks.
    if (!few) {
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        if (wave < ncols) {
            const int colv = col_of(wave);
            const int c = (kp + colv - 1) / colv;
            if (L <= 256) column_tau<4>(s_col + (size_t)wave * L, L, c, s_tau);
            else column_tau<8>(s_col + (size_t)wave * L, L, c, s_tau);
        }
        __syncthreads();
    }
    SEL_STAMP(2);
    const key_t64 tau = *s_tau;   // (few: still KEY_PAD -- every real key passes)

    // compact the survivors (keys <= tau) straight from the registers
#pragma unroll
    for (int i = 0; i < KREG; ++i) {
        if (i < n_round) {
            const key_t64 key = kreg[i];
            if (key != KEY_PAD && key <= tau) {
                const unsigned int slot = atomicAdd(&s_cnt[0], 1u);
                if (slot < (unsigned)SEL_SURV_CAP) s_surv[slot] = key;
            }
        }
    }
    __syncthreads();
    SEL_STAMP(3);
    const int S = min((int)s_cnt[0], SEL_SURV_CAP);  // bound above guarantees S <= cap for kp <= 72
    if (S <= 96) {
        // pair-parallel ranks: thread <-> (i, j), rank[i] += key[j] < key[i]
        for (int t = threadIdx.x; t < S; t += blockDim.x) s_srank[t] = 0;
        __syncthreads();
        for (int pr = threadIdx.x; pr < S * S; pr += blockDim.x) {
            const int i = pr / S, j = pr - i * S;
            if (s_surv[j] < s_surv[i]) atomicAdd(&s_srank[i], 1u);
        }
        __syncthreads();
        for (int t = threadIdx.x; t < S; t += blockDim.x)
            if ((int)s_srank[t] < kp) s_best[s_srank[t]] = s_surv[t];
    } else {
        // T = 1024 / S' threads share a survivor (S' = the power of two >= S): each ranks it against a T-th of the set and adds its
        // share to the survivor's counter.  (One thread per survivor walked all S keys alone while the other 1024 - S threads
        // idled: 176 survivors -- 16 lists x 11 keys, no column bounds -- took 2.9 us, 349 took 6.3.)  The counters reuse the column
        // entries' space, which nobody reads any more.
        unsigned int *s_rk = reinterpret_cast<unsigned int *>(s_col);
        for (int t = threadIdx.x; t < S; t += blockDim.x) s_rk[t] = 0;
        __syncthreads();
        int Sp = 128;
        while (Sp < S) Sp <<= 1;                      // <= SEL_SURV_CAP = SEL_THREADS
        const int T = SEL_THREADS / Sp;               // 8, 4, 2 or 1 threads per survivor
        const int e = (int)threadIdx.x / T, part = (int)threadIdx.x - e * T;
        if (e < S) {
            const key_t64 key = s_surv[e];
            const int per = (S + T - 1) / T, lo = part * per, hi = min(S, lo + per);
            unsigned int rank = 0;
            int i = lo;
#pragma unroll 1
            for (; i + 4 <= hi; i += 4) {
                key_t64 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = s_surv[i + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) rank += (x[u] < key) ? 1u : 0u;
            }
            for (; i < hi; ++i) rank += (s_surv[i] < key) ? 1u : 0u;
            if (rank) atomicAdd(&s_rk[e], rank);
        }
        __syncthreads();
        for (int t = threadIdx.x; t < S; t += blockDim.x)
            if ((int)s_rk[t] < kp) s_best[s_rk[t]] = s_surv[t];
    }
    __syncthreads();
    if (p.dbg && blockIdx.x == 0 && threadIdx.x == 0) p.dbg[8] = (unsigned long long)S;
    }   // (L > 1)
    SEL_STAMP(4);

    // ---- exact rescoring (bit-identical to the oracle's index-order f64 sums, see exact_sums()).
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    double my_ab = 0.0, my_b2 = 0.0;
    const key_t64 my_key = (int)threadIdx.x < kp ? s_best[threadIdx.x] : KEY_PAD;
    if (fast_rescore) {
        // The product of two f32 values is EXACT in f64, so "acc = acc + (double)a * (double)b" only rounds in
        // the add.  All 1024 threads compute the products q_i*c_i, c_i*c_i, q_i*q_i into LDS in parallel (one
        // wave per candidate row, coalesced 16 B per lane); what stays sequential is one chain of 256 dependent
        // f64 ADDS per sum, run by 2*k'+1 threads in three different waves.  (Before: each of the k' threads
        // converted and multiplied inside its chain -- 9.2k cycles for this stage, now the adds alone.)
        const f32x4 qv = reinterpret_cast<const f32x4 *>(p.queries + (size_t)qi * 256)[lane];
        // a wave owns candidates wave, wave+16, (wave+32): request all its rows first (ONE global latency)
        constexpr int ROWS_PER_WAVE = (SEL_FAST_KP + SEL_THREADS / 64 - 1) / (SEL_THREADS / 64);  // 3
        f32x4 bv[ROWS_PER_WAVE];
        bool have[ROWS_PER_WAVE];
#pragma unroll
        for (int u = 0; u < ROWS_PER_WAVE; ++u) {
            const int c = wave + u * n_waves;
            have[u] = c < kp && s_best[c] != KEY_PAD;
            const uint32_t row = have[u] ? (uint32_t)(s_best[c] & 0xFFFFFFFFull) : 0u;
            bv[u] = reinterpret_cast<const f32x4 *>(p.corpus + (uint64_t)row * 256)[lane];
        }
        // (conversions only now: the query and all row loads above are in flight together)
        const double q0 = qv.x, q1 = qv.y, q2 = qv.z, q3 = qv.w;
        if (wave == n_waves - 1) {
            double *dst = s_Q + 4 * lane;
            dst[0] = q0 * q0; dst[1] = q1 * q1; dst[2] = q2 * q2; dst[3] = q3 * q3;
            // the query's largest magnitude, for the domain verdict below (domain.hip)
            atomicMax(&s_cnt[2], max(max(__float_as_uint(qv.x) & 0x7fffffffu, __float_as_uint(qv.y) & 0x7fffffffu),
                                     max(__float_as_uint(qv.z) & 0x7fffffffu, __float_as_uint(qv.w) & 0x7fffffffu)));
        }
#pragma unroll
        for (int u = 0; u < ROWS_PER_WAVE; ++u) {
            if (have[u]) {
                const int c = wave + u * n_waves;
                const double b0 = bv[u].x, b1 = bv[u].y, b2 = bv[u].z, b3 = bv[u].w;
                double *dp = s_P + (size_t)c * PROD_STRIDE + 4 * lane;
                double *db = s_B + (size_t)c * PROD_STRIDE + 4 * lane;
                dp[0] = q0 * b0; dp[1] = q1 * b1; dp[2] = q2 * b2; dp[3] = q3 * b3;
                db[0] = b0 * b0; db[1] = b1 * b1; db[2] = b2 * b2; db[3] = b3 * b3;
            }
        }
        __syncthreads();
        SEL_STAMP(5);
        // chains: threads [0,kp) sum P, threads [64,64+kp) sum B, thread 128 sums Q
        const int role = (int)threadIdx.x >> 6, idx = (int)threadIdx.x & 63;
        const double *src = nullptr;
        if (role == 0 && idx < kp && s_best[idx] != KEY_PAD) src = s_P + (size_t)idx * PROD_STRIDE;
        else if (role == 1 && idx < kp && s_best[idx] != KEY_PAD) src = s_B + (size_t)idx * PROD_STRIDE;
        else if (threadIdx.x == 128) src = s_Q;
        if (src) {
            double acc = 0.0;
#pragma unroll 16
            for (int i = 0; i < 256; ++i) acc = acc + src[i];
            if (role == 0) my_ab = acc;
            else if (role == 1) s_b2[idx] = acc;
            else *s_a2 = acc;
        }
        __syncthreads();
        if ((int)threadIdx.x < kp) my_b2 = s_b2[threadIdx.x];
    } else {
        // large k': stage the candidate rows (one wave per row, 16 B per lane) and the query converted to f64
        if (threadIdx.x < 256) {
            const float qf = p.queries[(size_t)qi * 256 + threadIdx.x];
            s_qd[threadIdx.x] = (double)qf;
            atomicMax(&s_cnt[2], __float_as_uint(qf) & 0x7fffffffu);
        }
        for (int c = wave; c < kp; c += n_waves) {
            if (s_best[c] != KEY_PAD) {
                const float *g = p.corpus + (uint64_t)(uint32_t)(s_best[c] & 0xFFFFFFFFull) * 256;
                s_rows[(size_t)c * ROW_STRIDE_F4 + lane] = reinterpret_cast<const f32x4 *>(g)[lane];
            }
        }
        __syncthreads();
        SEL_STAMP(5);
        // thread t < kp walks candidate t's row; thread 128 (another wave) walks the query
        if (my_key != KEY_PAD) {
            const f32x4 *r4 = s_rows + (size_t)threadIdx.x * ROW_STRIDE_F4;
#pragma unroll 8
            for (int i = 0; i < 64; ++i) {
                const f32x4 b = r4[i];
                const double a0 = s_qd[4 * i], a1 = s_qd[4 * i + 1], a2 = s_qd[4 * i + 2], a3 = s_qd[4 * i + 3];
                const double bx = b.x, by = b.y, bz = b.z, bw = b.w;
                my_ab = my_ab + a0 * bx; my_b2 = my_b2 + bx * bx;
                my_ab = my_ab + a1 * by; my_b2 = my_b2 + by * by;
                my_ab = my_ab + a2 * bz; my_b2 = my_b2 + bz * bz;
                my_ab = my_ab + a3 * bw; my_b2 = my_b2 + bw * bw;
            }
        }
        if (threadIdx.x == 128) {
            double a2 = 0.0;
#pragma unroll 8
            for (int i = 0; i < 256; ++i) a2 = a2 + s_qd[i] * s_qd[i];
            *s_a2 = a2;
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < kp) {
        double d = __builtin_inf();
        uint32_t r = 0xFFFFFFFFu;
        if (my_key != KEY_PAD) {
            r = (uint32_t)(my_key & 0xFFFFFFFFull);
            d = cos_finish_exact(my_ab, *s_a2, my_b2);
            if (p.ws_threshold) {
                // qdrant score_threshold keeps score > threshold (similarity metrics)
                if (!((1.0 - d) > (double)p.ws_thr_score)) { d = __builtin_inf(); r = 0xFFFFFFFFu; }
            }
        }
        s_d[threadIdx.x] = d;
        s_r[threadIdx.x] = r;
        if (r != 0xFFFFFFFFu) atomicAdd(&s_cnt[1], 1u);
    }
    uint64_t *orow = p.out_rows + (size_t)qi * p.out_stride;
    double *odist = p.out_dist + (size_t)qi * p.out_stride;
    if (threadIdx.x < p.k_out) { orow[threadIdx.x] = 0xFFFFFFFFFFFFFFFFull; odist[threadIdx.x] = __builtin_inf(); }
    if (threadIdx.x == 0 && p.out_uncertain) p.out_uncertain[qi] = 0;
    if (threadIdx.x == 0 && p.out_status) p.out_status[qi] = 0;
    __syncthreads();
    SEL_STAMP(6);
    for (int pr = threadIdx.x; pr < kp * kp; pr += blockDim.x) {
        const int i = pr / kp, j = pr - i * kp;
        const uint32_t ri = s_r[i], rj = s_r[j];
        if (ri != 0xFFFFFFFFu && rj != 0xFFFFFFFFu) {
            const double di = s_d[i], dj = s_d[j];
            if (dj < di || (dj == di && rj < ri)) atomicAdd(&s_rank[i], 1u);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < kp && s_r[threadIdx.x] != 0xFFFFFFFFu) {
        const unsigned rank = s_rank[threadIdx.x];
        if (rank < p.k_out) { orow[rank] = p.row_base + s_r[threadIdx.x]; odist[rank] = s_d[threadIdx.x]; }
    }
    if (!magnitude_in_domain(s_cnt[2])) {
        // The query is outside the library's domain (a non-finite component, or a magnitude the f32 nominating kernels do not answer
        // for: domain.hip).  The host entry points refuse such queries before anything is launched; a device entry point says so
        // here -- SMT_STATUS_INVALID_QUERY: the list means nothing.
        if (threadIdx.x == 0) {
            if (p.out_uncertain) p.out_uncertain[qi] = 3;
            if (p.out_status) p.out_status[qi] = 3u;
            if (p.status) atomicAdd(p.status, 1ull);
        }
    } else if (p.f32_err > 0.0 && s_best[kp - 1] != KEY_PAD) {
        // Exactness certificate (SelectArgs::f32_err).  kp rows were nominated, so rows outside the lists may
        // exist; each of them has f32 distance >= tau32, hence exact distance >= tau32 - f32_err.  Exactly one
        // thread decides: the owner of the k_out-th result, or thread 0 when fewer than k_out rows qualified
        // (only possible with the workspace score threshold, which then bounds what an outside row would need).
        const double floor_out = (double)__uint_as_float((unsigned)(s_best[kp - 1] >> 32)) - p.f32_err;
        bool decide = false, uncertain = false;
        if ((int)threadIdx.x < kp && s_r[threadIdx.x] != 0xFFFFFFFFu && s_rank[threadIdx.x] == p.k_out - 1) {
            decide = true;
            uncertain = !(floor_out > s_d[threadIdx.x]);
        } else if (threadIdx.x == 0 && s_cnt[1] < p.k_out) {
            decide = true;
            uncertain = p.ws_threshold ? ((1.0 - floor_out) > (double)p.ws_thr_score) : true;
        }
        unsigned int code = uncertain ? 1u : 0u;   // SMT_STATUS_UNCERTAIN
        if constexpr (OVF) {
            if (decide && p.overflow[qi]) { uncertain = true; code = 2u; }   // rows were dropped on the way (SelectArgs::overflow): SMT_STATUS_OVERFLOW
        }
        if (decide && uncertain) {
            if (p.out_uncertain) p.out_uncertain[qi] = code;
            if (p.out_status) p.out_status[qi] = code;
            if (p.status) atomicAdd(p.status, 1ull);
        }
    } else if constexpr (OVF) {
        if (threadIdx.x == 0 && p.overflow[qi]) {
            if (p.out_uncertain) p.out_uncertain[qi] = 2;
            if (p.out_status) p.out_status[qi] = 2u;
            if (p.status) atomicAdd(p.status, 1ull);
        }
    }
    if (threadIdx.x == 0 && p.out_counts)
        p.out_counts[qi] = s_cnt[1] < p.k_out ? s_cnt[1] : p.k_out;
    if (p.flags) {
        __syncthreads();
        if (threadIdx.x == 0 && blockIdx.x == 0)  // async mode: one query per launch
            __hip_atomic_store(p.flags + 1, p.step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (p.host_flag) {
        // Delivery (common.h): count this block as finished -- thread 0's agent-scope release covers the block's output stores, ordered
        // before it by the barrier, the same hand-over the scan kernel uses for its lists -- and let the LAST block carry every query's
        // answer home: ONE wave copies the device block to pinned host memory and raises the completion word behind it (its own
        // stores, one system-scope release; a fence in each of the 16 waves would be 16 write-backs of the L2).
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long prev = __hip_atomic_fetch_add(p.done, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_cnt[3] = prev == (unsigned long long)gridDim.x - 1ull ? 1u : 0u;
        }
        __syncthreads();
        if (s_cnt[3] && threadIdx.x < 64) {
            for (uint32_t i = threadIdx.x; i < p.out_words; i += 64)
                p.host_out[i] = __hip_atomic_load(p.dev_out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.done, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: the copy is out before the flag
            if (threadIdx.x == 0) __hip_atomic_store(p.host_flag, p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    SEL_STAMP(7);
}

template <int KREG, bool OVF = false>
__global__ void __launch_bounds__(SEL_THREADS) final_select_kernel(FinalParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    final_select_body<KREG, OVF>(p, blockIdx.x, smem_raw);
}

static size_t final_smem_bytes(uint32_t n_lists, uint32_t kp)
{
    (void)n_lists;
    const size_t kp2 = (kp + 1) & ~1u;
    const size_t small = (size_t)(96 + 80) * 4 + 256 * 8 + kp2 * 8 * 3 + (size_t)((kp + 3) & ~3u) * 4 + 8 + 8 + 16;
    const size_t big_select = (size_t)(kp + 1) * ROW_STRIDE_F4 * 16 + (size_t)SEL_SURV_CAP * 8 + (size_t)SEL_MAX_COLS * SEL_MAX_LISTS * 8;
    const size_t big_rescore = kp <= (uint32_t)SEL_FAST_KP ? ((size_t)2 * kp * PROD_STRIDE + 256) * 8 : 0;
    return small + std::max(big_select, big_rescore) + 64;
}

// ------------------------------------------------- cross-shard top-k merge
// Entry (list l, query qi, slot i): rows[l*list_stride + qi*query_stride + i], same for dist.
// Plain layout [n_lists][nq][k_in]: list_stride = nq*k_in, query_stride = k_in.  Packed layout
// [n_lists][nq][2][k_in] (rows then distance bits, one all-gather instead of two):
// list_stride = nq*2*k_in, query_stride = 2*k_in, dist = rows + k_in.  One block per query.
__global__ void merge_topk_kernel(const uint64_t *rows, const double *dist, uint32_t n_lists, uint64_t list_stride,
                                  uint64_t query_stride, uint32_t k_in, uint32_t k_out, uint64_t *out_rows,
                                  double *out_dist, uint64_t out_query_stride)
{
    const uint32_t qi = blockIdx.x;
    const uint32_t M = n_lists * k_in;
    uint64_t *orow = out_rows + (size_t)qi * out_query_stride;
    double *odist = out_dist + (size_t)qi * out_query_stride;
    for (uint32_t t = threadIdx.x; t < k_out; t += blockDim.x) {
        orow[t] = 0xFFFFFFFFFFFFFFFFull;
        odist[t] = __builtin_inf();
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < M; e += blockDim.x) {
        const size_t idx = (size_t)(e / k_in) * list_stride + (size_t)qi * query_stride + (e % k_in);
        const uint64_t r = rows[idx];
        if (r == 0xFFFFFFFFFFFFFFFFull) continue;
        const double d = dist[idx];
        uint32_t rank = 0;
        for (uint32_t f = 0; f < M; ++f) {
            const size_t jdx = (size_t)(f / k_in) * list_stride + (size_t)qi * query_stride + (f % k_in);
            const uint64_t rf = rows[jdx];
            if (rf == 0xFFFFFFFFFFFFFFFFull) continue;
            const double df = dist[jdx];
            if (df < d || (df == d && rf < r)) ++rank;
        }
        if (rank < k_out) { orow[rank] = r; odist[rank] = d; }
    }
}

// The same merge over lists that are NOT gathered: list l starts at src.list[l] -- for a one-process group the exchange buffer of
// rank l, read IN PLACE over xGMI (peer access) or on this device (logical ranks).  Nothing is copied between the ranks: the
// exchange is this kernel's loads (n_lists x k_in x 16 B per query -- 1280 B at 8 ranks, k = 10: latency-bound), ordered
// after the ranks' select kernels by one stream event per rank (group.cpp: peer transport).  The candidates are staged in LDS once
// (16 B each, M <= 4096) so that the M x M rank computation never goes back over the links.
__global__ void merge_topk_sources_kernel(MergeSources src, uint32_t n_lists, uint64_t query_stride, uint32_t k_in, uint32_t k_out,
                                          uint64_t *out_rows, double *out_dist, uint64_t out_query_stride)
{
    extern __shared__ unsigned char smem_raw[];
    uint64_t *s_row = reinterpret_cast<uint64_t *>(smem_raw);
    const uint32_t qi = blockIdx.x;
    const uint32_t M = n_lists * k_in;
    double *s_dist = reinterpret_cast<double *>(s_row + M);
    uint64_t *orow = out_rows + (size_t)qi * out_query_stride;
    double *odist = out_dist + (size_t)qi * out_query_stride;
    for (uint32_t t = threadIdx.x; t < k_out; t += blockDim.x) {
        orow[t] = 0xFFFFFFFFFFFFFFFFull;
        odist[t] = __builtin_inf();
    }
    for (uint32_t e = threadIdx.x; e < M; e += blockDim.x) {
        const uint64_t *l = src.list[e / k_in] + (size_t)qi * query_stride;
        const uint32_t i = e % k_in;
        s_row[e] = l[i];
        s_dist[e] = reinterpret_cast<const double *>(l + k_in)[i];
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < M; e += blockDim.x) {
        const uint64_t r = s_row[e];
        if (r == 0xFFFFFFFFFFFFFFFFull) continue;
        const double d = s_dist[e];
        uint32_t rank = 0;
        for (uint32_t f = 0; f < M; ++f) {
            const uint64_t rf = s_row[f];
            if (rf == 0xFFFFFFFFFFFFFFFFull) continue;
            const double df = s_dist[f];
            if (df < d || (df == d && rf < r)) ++rank;
        }
        if (rank < k_out) { orow[rank] = r; odist[rank] = d; }
    }
}

// ------------------------------------------------------------------ launchers
static inline uint32_t candidates_per_list(const smt_ctx *ctx, uint32_t k_out)
{
    // guard band: 8 extra f32 candidates so that f32-vs-f64 rank flips at the k-th boundary do not drop a true
    // top-k row; the select stage PROVES per query that the band was wide enough (SelectArgs::f32_err) and flags
    // the query otherwise.  k_out <= SCAN_MAX_K = 56, so the band never shrinks below 8; tuning key guard_band widens
    // it (lists hold at most 64 keys) for corpora with many near-duplicate lines -- every proof that succeeds saves
    // the exhaustive re-answer, at the price of a longer select (256 lists x k' keys).
    return std::min<uint32_t>(64, k_out + (uint32_t)ctx->tune.guard_band);
}

template <int NQ, int U>
static int launch_scan_variant(smt_ctx *ctx, const ScanParams &p, int blocks, int threads, bool nt)
{
    const size_t smem = (size_t)NQ * (threads / 64) * 64 * sizeof(key_t64) + 16 + 64 + 256;  // per-query, per-wave key slots (the chunk counter and the STEAL ring live in the second query's until the merge) + the waves' key counts
    dim3 g(blocks), b(threads);
    if constexpr (U == 4) {
        if (p.steal_ctr) {   // groups of rounds dealt while the kernel runs (512-thread blocks, checked by the caller)
            if (nt) hipLaunchKernelGGL((scan_topk_kernel<NQ, U, true, false, true>), g, b, smem, ctx->stream, p);
            else hipLaunchKernelGGL((scan_topk_kernel<NQ, U, false, false, true>), g, b, smem, ctx->stream, p);
            SMT_HIP_CHECK(hipGetLastError());
            return SMT_OK;
        }
    }
    if (nt) hipLaunchKernelGGL((scan_topk_kernel<NQ, U, true, false>), g, b, smem, ctx->stream, p);
    else hipLaunchKernelGGL((scan_topk_kernel<NQ, U, false, false>), g, b, smem, ctx->stream, p);
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

template <int NQ>
static int launch_scan_filtered(smt_ctx *ctx, const ScanParams &p, int blocks, int threads, bool nt)
{
    const size_t smem = (size_t)NQ * (threads / 64) * 64 * sizeof(key_t64) + 16 + 64 + 256;
    dim3 g(blocks), b(threads);
    if (nt) hipLaunchKernelGGL((scan_topk_kernel<NQ, FILTER_CHUNK, true, true>), g, b, smem, ctx->stream, p);
    else hipLaunchKernelGGL((scan_topk_kernel<NQ, FILTER_CHUNK, false, true>), g, b, smem, ctx->stream, p);
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

// Block lists -> final answer in ONE launch (per query: prune + rank + rescore).
int launch_select(smt_ctx *ctx, const SelectArgs &a)
{
    SMT_REQUIRE(a.n_lists >= 1 && a.n_lists <= (uint32_t)SEL_MAX_LISTS, "select stage accepts 1..512 block lists");
    SMT_REQUIRE(a.async_step == 0 || (a.nq == 1 && ctx->aux_stream && ctx->d_flags), "async select handles one query per launch");
    if (!(ctx->attr_done & ATTR_SELECT)) {  // per context == per device (a group runs several GPUs in one process)
        SMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(final_select_kernel<8>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(final_select_kernel<36>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(final_select_kernel<8, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(final_select_kernel<36, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->attr_done |= ATTR_SELECT;
    }
    FinalParams f;
    f.corpus = a.corpus;
    f.queries = a.queries;
    f.lists = a.lists;
    f.list_stride = a.list_stride;
    f.n_lists = a.n_lists;
    f.kp = a.kp;
    f.k_out = a.k_out;
    f.ws_threshold = a.ws_threshold;
    f.ws_thr_score = a.ws_thr_score;
    f.row_base = a.row_base;
    f.out_rows = a.out_rows;
    f.out_dist = a.out_dist;
    f.out_counts = a.out_counts;
    f.out_stride = a.out_stride ? a.out_stride : a.k_out;
    f.f32_err = a.f32_err;
    f.out_uncertain = a.out_uncertain;
    f.out_status = a.out_status;
    f.status = ctx->d_status;
    f.dbg = reinterpret_cast<unsigned long long *>(ctx->tune.select_debug_ptr);
    f.flags = a.async_step ? ctx->d_flags : nullptr;
    f.step = a.async_step;
    f.overflow = a.overflow;
    const Delivery *dl = a.async_step ? nullptr : a.deliver;
    f.dev_out = dl ? dl->dev_out : nullptr;
    f.host_out = dl ? dl->host_out : nullptr;
    f.host_flag = dl ? dl->host_flag : nullptr;
    f.done = dl ? dl->done : nullptr;
    f.seq = dl ? dl->seq : 0;
    f.out_words = dl ? dl->n_words : 0;
    const bool small = (uint64_t)a.n_lists * a.kp <= (uint64_t)8 * SEL_THREADS;
    const size_t smem = final_smem_bytes(a.n_lists, a.kp);
    if (a.async_step) {
        if (small) hipLaunchKernelGGL(final_select_kernel<8>, dim3(a.nq), dim3(SEL_THREADS), smem, ctx->aux_stream, f);
        else hipLaunchKernelGGL(final_select_kernel<36>, dim3(a.nq), dim3(SEL_THREADS), smem, ctx->aux_stream, f);
        ctx->async_pending = true;
        SMT_HIP_CHECK(hipGetLastError());
        return SMT_OK;
    }
    if (ctx->tune.prof_select) prof_begin(ctx, "select");
    if (a.overflow) {
        if (small) hipLaunchKernelGGL((final_select_kernel<8, true>), dim3(a.nq), dim3(SEL_THREADS), smem, ctx->stream, f);
        else hipLaunchKernelGGL((final_select_kernel<36, true>), dim3(a.nq), dim3(SEL_THREADS), smem, ctx->stream, f);
    } else {
        if (small) hipLaunchKernelGGL(final_select_kernel<8>, dim3(a.nq), dim3(SEL_THREADS), smem, ctx->stream, f);
        else hipLaunchKernelGGL(final_select_kernel<36>, dim3(a.nq), dim3(SEL_THREADS), smem, ctx->stream, f);
    }
    if (ctx->tune.prof_select) prof_end(ctx, "select");
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

int launch_scan_topk(smt_ctx *ctx, const ScanArgs &a)
{
    SMT_REQUIRE(a.k_out >= 1 && a.k_out <= SCAN_MAX_K, "top_k for the scan path must be in [1, 56]");
    SMT_REQUIRE(a.rows < 0xFFFFFFFFull, "a shard holds fewer than 2^32-1 rows");
    const bool filtered = a.n_ranges > 0;
    const uint32_t kp = candidates_per_list(ctx, a.k_out);
    int blocks = ctx->tune.scan_blocks > 0 ? ctx->tune.scan_blocks : ctx->num_cus;
    if (blocks > SEL_MAX_LISTS) blocks = SEL_MAX_LISTS;
    const int threads = ctx->tune.scan_threads;
    const int U = filtered ? FILTER_CHUNK : ctx->tune.scan_unroll;
    const uint64_t n_chunks = filtered ? a.n_chunks : (a.n_virtual + (uint64_t)U - 1) / (uint64_t)U;
    const uint64_t waves_per_block = (uint64_t)(threads / 64);
    if ((uint64_t)blocks * waves_per_block > n_chunks) {  // tiny inputs: do not launch idle blocks
        const uint64_t need = (n_chunks + waves_per_block - 1) / waves_per_block;
        blocks = (int)(need > 0 ? need : 1);
    }
    // async select (one query per call): two list buffers alternate, the select of step i reads buffer i&1 on
    // the aux stream while the scan of step i+1 fills the other one
    const bool async = a.allow_async && ctx->tune.async_select && a.nq == 1;
    int rc = SMT_OK;
    if (async && (rc = ensure_async(ctx))) return rc;
    if (!async && (rc = drain_async(ctx))) return rc;
    const uint64_t step = async ? ++ctx->async_step : 0;
    const size_t list_bytes = (((size_t)a.nq * blocks * kp * sizeof(key_t64)) + 255) & ~(size_t)255;
    const size_t table_bytes = filtered ? (size_t)n_chunks * sizeof(uint64_t) : 0;
    rc = ensure_scratch(ctx, 2 * list_bytes + table_bytes);
    if (rc != SMT_OK) return rc;
    key_t64 *lists = reinterpret_cast<key_t64 *>(reinterpret_cast<char *>(ctx->d_scratch) + (step & 1) * list_bytes);
    uint64_t *table = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(ctx->d_scratch) + 2 * list_bytes);

    const bool nt = ctx->tune.scan_nontemporal != 0;
    prof_begin(ctx, "scan");
    const uint64_t *use_table = table;
    if (filtered && (rc = range_chunk_table(ctx, a, table, &use_table))) return rc;
    for (uint32_t q0 = 0; q0 < a.nq;) {
        ScanParams p;
        p.corpus = a.corpus;
        p.queries = a.queries + (size_t)q0 * 256;
        p.n_virtual = a.n_virtual;
        p.chunk_table = filtered ? use_table : nullptr;
        p.n_chunks = n_chunks;
        p.kp = kp;
        p.block_lists = lists + (size_t)q0 * blocks * kp;
        p.stamps = reinterpret_cast<unsigned long long *>(ctx->tune.scan_debug_ptr);
        p.flags = async ? ctx->d_flags : nullptr;
        p.step = step;
        p.steal_ctr = nullptr;
        p.steal_lr = 0;
        // Dynamic groups (scan_topk_kernel STEAL; tuning key scan_steal = rounds per group, 0 = the static deal): unfiltered scans of
        // 8-wave blocks over enough rows that every block gets well past its static groups.  Each launch has its own counter: a ring
        // of 64, zeroed together every 64th launch (one 256-byte memset in stream order).
        if (!filtered && U == 4 && threads == 512 && ctx->tune.scan_steal > 0 &&
            n_chunks >= (uint64_t)blocks * 8u * (uint64_t)ctx->tune.scan_steal * 8u) {
            if (!ctx->d_steal) SMT_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_steal), 64 * sizeof(unsigned int)));
            if (ctx->steal_seq % 64 == 0) SMT_HIP_CHECK(hipMemsetAsync(ctx->d_steal, 0, 64 * sizeof(unsigned int), ctx->stream));
            p.steal_ctr = ctx->d_steal + (ctx->steal_seq % 64);
            ++ctx->steal_seq;
            uint32_t lr = 0;
            while ((2 << lr) <= ctx->tune.scan_steal) ++lr;
            p.steal_lr = lr;
            // the dynamic share: scan_steal_pct per cent of the groups (at least STEAL_DEPTH + 1 per block stay static)
            const uint64_t n_groups = (n_chunks + (8ull << lr) - 1) / (8ull << lr);
            const uint64_t per_block = n_groups / (uint64_t)blocks;
            const uint64_t n_static = per_block * (uint64_t)(100 - ctx->tune.scan_steal_pct) / 100;
            p.steal_static = (uint32_t)std::max<uint64_t>(n_static, 3);
        }
        const uint32_t left = a.nq - q0;
        p.nq_active = left >= 4 ? 4u : left;
        // (three queries ride in the four-query kernel: with the four rows of a chunk reduced together a pass costs 1.06 x / 1.4 x one
        // query's for two / four queries -- 172 / 230 us at 1 M rows -- against 335 us for a pass of two and a pass of one)
        if (left >= 3) {
            rc = filtered ? launch_scan_filtered<4>(ctx, p, blocks, threads, nt)
                 : (U == 4) ? launch_scan_variant<4, 4>(ctx, p, blocks, threads, nt)
                            : launch_scan_variant<4, 8>(ctx, p, blocks, threads, nt);
            q0 += p.nq_active;
        } else if (left >= 2) {
            rc = filtered ? launch_scan_filtered<2>(ctx, p, blocks, threads, nt)
                 : (U == 4) ? launch_scan_variant<2, 4>(ctx, p, blocks, threads, nt)
                            : launch_scan_variant<2, 8>(ctx, p, blocks, threads, nt);
            q0 += 2;
        } else {
            if (filtered) rc = launch_scan_filtered<1>(ctx, p, blocks, threads, nt);
            else if (U == 2) rc = launch_scan_variant<1, 2>(ctx, p, blocks, threads, nt);
            else if (U == 4) rc = launch_scan_variant<1, 4>(ctx, p, blocks, threads, nt);
            else rc = launch_scan_variant<1, 8>(ctx, p, blocks, threads, nt);
            q0 += 1;
        }
        if (rc != SMT_OK) return rc;
    }
    prof_end(ctx, "scan");
    SelectArgs s;
    s.corpus = a.corpus;
    s.queries = a.queries;
    s.nq = a.nq;
    s.lists = lists;
    s.n_lists = (uint32_t)blocks;
    s.kp = kp;
    s.list_stride = (uint64_t)blocks * kp;
    s.k_out = a.k_out;
    s.ws_threshold = a.ws_threshold;
    s.ws_thr_score = a.ws_thr_score;
    s.row_base = a.row_base;
    s.out_rows = a.out_rows;
    s.out_dist = a.out_dist;
    s.out_counts = a.out_counts;
    s.async_step = step;
    s.out_stride = a.out_stride;
    s.f32_err = F32_ERR_SCAN;
    s.out_uncertain = a.out_uncertain;
    s.out_status = a.out_status;
    s.deliver = a.deliver;
    return launch_select(ctx, s);
}

int launch_rescore_rows(smt_ctx *ctx, const float *corpus, const float *query, const uint32_t *rows,
                        uint64_t n, double *out_dist)
{
    if (n == 0) return SMT_OK;
    const int threads = 64;
    const uint64_t blocks = (n + threads - 1) / threads;
    prof_begin(ctx, "select");
    hipLaunchKernelGGL(rescore_rows_kernel, dim3((unsigned)blocks), dim3(threads), 0, ctx->stream, corpus,
                       query, rows, n, out_dist);
    prof_end(ctx, "select");
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

int launch_gather_rows256(smt_ctx *ctx, const float *src, const uint32_t *idx_dev, uint32_t n, float *dst)
{
    if (n == 0) return SMT_OK;
    hipLaunchKernelGGL(gather_rows256_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, src, idx_dev, n, dst);
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

int launch_rescore_rows_multi(smt_ctx *ctx, const float *corpus, const float *queries, const uint32_t *rows, const uint32_t *qidx,
                              uint64_t n, double *out_dist)
{
    if (n == 0) return SMT_OK;
    const int threads = 64;
    prof_begin(ctx, "select");
    hipLaunchKernelGGL(rescore_rows_multi_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0, ctx->stream, corpus,
                       queries, rows, qidx, n, out_dist);
    prof_end(ctx, "select");
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

int launch_merge_topk(smt_ctx *ctx, const uint64_t *rows, const double *dist, uint32_t n_lists,
                      uint32_t nq, uint32_t k_in, uint32_t k_out, uint64_t *out_rows, double *out_dist)
{
    SMT_REQUIRE((uint64_t)n_lists * k_in <= 8192, "device merge handles up to 8192 candidates per query");
    hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(256), 0, ctx->stream, rows, dist, n_lists, (uint64_t)nq * k_in,
                       (uint64_t)k_in, k_in, k_out, out_rows, out_dist, (uint64_t)k_out);
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

// Packed lists [n_lists][nq][2][k_in] (list_stride_words apart: the exchange buffers of group.cpp carry a few status
// words behind each rank's lists; 0 = dense) -> out_packed [nq][2][k_out], on stream `st`.  n_lists == 0 writes padding.
int launch_merge_topk_packed_on(smt_ctx *ctx, hipStream_t st, const uint64_t *packed, uint32_t n_lists, uint32_t nq, uint32_t k_in,
                                uint32_t k_out, uint64_t *out_packed, uint64_t list_stride_words)
{
    (void)ctx;
    SMT_REQUIRE((uint64_t)n_lists * k_in <= 8192, "device merge handles up to 8192 candidates per query");
    const uint64_t ls = list_stride_words ? list_stride_words : (uint64_t)nq * 2 * k_in;
    hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(256), 0, st, packed, reinterpret_cast<const double *>(packed + k_in),
                       n_lists, ls, (uint64_t)2 * k_in, k_in, k_out, out_packed, reinterpret_cast<double *>(out_packed + k_out),
                       (uint64_t)2 * k_out);
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

// Packed lists [nq][2][k_in], one per source pointer (see merge_topk_sources_kernel) -> out_packed [nq][2][k_out] on stream `st`.
int launch_merge_topk_sources_on(hipStream_t st, const MergeSources &src, uint32_t n_lists, uint32_t nq, uint32_t k_in, uint32_t k_out,
                                 uint64_t *out_packed)
{
    SMT_REQUIRE(n_lists >= 1 && n_lists <= SMT_MAX_MERGE_SOURCES, "1..64 lists are merged in place");
    SMT_REQUIRE((uint64_t)n_lists * k_in <= 4096, "the in-place merge stages up to 4096 candidates per query");
    const size_t smem = (size_t)n_lists * k_in * 16;
    hipLaunchKernelGGL(merge_topk_sources_kernel, dim3(nq), dim3(256), smem, st, src, n_lists, (uint64_t)2 * k_in, k_in, k_out, out_packed,
                       reinterpret_cast<double *>(out_packed + k_out), (uint64_t)2 * k_out);
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

__global__ void combine_status_kernel(MergeSources src, const uint64_t *base, uint64_t stride_words, uint32_t n, uint32_t nq, uint32_t *out)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    uint64_t worst = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const uint64_t v = base ? base[(size_t)j * stride_words + q] : src.list[j][q];
        worst = v > worst ? v : worst;
    }
    out[q] = (uint32_t)worst;
}

int launch_combine_status_on(hipStream_t st, const MergeSources *src, const uint64_t *base, uint64_t stride_words, uint32_t n, uint32_t nq,
                             uint32_t *out)
{
    SMT_REQUIRE((src != nullptr) != (base != nullptr), "status sources: pointers or a strided buffer");
    SMT_REQUIRE(src == nullptr || n <= SMT_MAX_MERGE_SOURCES, "1..64 status sources are read in place");
    if (nq == 0) return SMT_OK;
    MergeSources none{};
    hipLaunchKernelGGL(combine_status_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, src ? *src : none, base, stride_words, n, nq, out);
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

int launch_merge_topk_packed(smt_ctx *ctx, const uint64_t *packed, uint32_t n_lists, uint32_t nq, uint32_t k_in,
                             uint32_t k_out, uint64_t *out_packed)
{
    // async pipeline: the merge follows the select it consumes on the aux stream (tuning key merge_on_aux)
    const bool on_aux = ctx->tune.merge_on_aux && ctx->aux_stream != nullptr;
    hipStream_t st = on_aux ? ctx->aux_stream : ctx->stream;  // (ctx->stream may itself be the null stream)
    if (on_aux) ctx->async_pending = true;
    return launch_merge_topk_packed_on(ctx, st, packed, n_lists, nq, k_in, k_out, out_packed, 0);
}

}  // namespace smt
