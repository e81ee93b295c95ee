// gemm.h -- what the K3 translation units share: the candidate buffers and the level plan, the parameters of a level launch, the
// epilogue that turns an accumulator tile into nominations, the geometry constants the host side plans with, and the launchers
// each kernel family exports (a family's kernels and their launches live in ONE translation unit: gemm_level.hip,
// gemm_rowreg.hip, gemm_ldsrow.hip; gemm_topk.hip plans the levels and owns the small per-batch kernels).  DESIGN.md 4.3.
#pragma once
#include <type_traits>

#include "common.h"
#include "device_utils.h"
#include "mfma_tile.h"

namespace smt {

constexpr uint32_t CAND_CAP = 2048;            // candidate slots per query
// Level plan: tiles are visited in levels of geometrically growing size (every ratio^j-th tile first).  Level 0
// (<= LEVEL0_MAX_TILES tiles = 1024 rows) appends every row; each later level appends about ratio x k' candidates
// per query.  Measured alternative (MI355X, 10 M rows): ratio 64 with 4096 slots -- three levels instead of five --
// LOSES: its middle level (4.8 k tiles) leaves every wave two tiles, the O(n^2) level select over ~3 k candidates
// costs 90 us per level instead of 8 (32 queries: 2.34 vs 2.07 ms per batch; 1000 queries: 41.2 vs 39.8 ms).
// Round 3, with the operand image: ratios 8 and 4 (one resp. three more levels, half resp. a quarter of the nominations per
// level) measure within +-2 % of 16 from 1 to 1000 queries (1000 x 10 M: 6.15 / 6.17 against 6.25 ms per call).
constexpr int LEVEL_RATIO_SMALL_K = 16;
constexpr int LEVEL_RATIO_LARGE_K = 16;
constexpr uint32_t LEVEL_RATIO_KP_LIMIT = 24;
constexpr int LEVEL0_MAX_TILES = 32;
// gemm_rowreg_kernel starts from a BOOTSTRAP level instead: up to this many tiles (every stride-th), no candidate lists, only the
// tiles' best distances -- the kp-th smallest of them bounds the final kp-th distance (the tiles are distinct, so are their best rows)
// and is the first threshold.  Three levels (5, 77, 1221 tiles at 10 M rows: three launches whose nominations are slot grabs on the
// same thousand counters with a memory round trip each, and three select passes) became one pass of plain stores.
constexpr uint32_t BOOTSTRAP_MAX_TILES = 8192;   // (their minima are folded into 1024 slots: up to 8 tiles per slot)
constexpr uint32_t BOOTSTRAP_SLOTS = 1024;
constexpr uint32_t GEMM_MAX_NQ = 3584;         // 4 x 32.5 KiB tile slots + 8 B per query fit the 160 KiB LDS

struct GemmParams {
    const float *corpus;
    uint64_t n_rows;
    const float *queries;     // [nq][256]
    const uint32_t *queries_split;  // BF16 kernels: [nqt*32][256] words, the split image written by split_queries_kernel
    uint32_t nq;
    uint32_t nqt;             // ceil(nq / 32)
    uint64_t level_tiles;     // tiles of the level: this launch visits [tile_begin, level_tiles) (gemm_rowreg_kernel; the others start at 0)
    uint64_t tile_begin;
    uint64_t stride;          // visited tile = stride * u(i)
    int skip16;               // 64, 16 or 4 when u skips the multiples of that ratio (they belong to earlier levels), else 0
    uint32_t qsplit;          // gemm_level_kernel: blocks per row-tile group, each sweeping 1/qsplit of the query tiles
    const float *tau;         // [nqt*32] distance thresholds (+inf = take everything, <0 = padding)
    const float *qconst;      // gemm_rowreg_kernel: [nqt*32][2] = (score threshold = score_threshold(tau, rq), 1/|q|) per query;
                              // the thresholds are kept by level_select_kernel; padding queries: (-1, 0)
    key_t64 *cand;            // [nq][CAND_CAP]
    unsigned int *counts;     // [nq]
    // range-filtered batches (gemm_ldsrow_kernel<.., true>): the rows to scan are the FILTER_CHUNK-row chunks of the
    // chunk table (scan_kernels.hip: row0 | valid rows << 32); a "tile" is then 8 consecutive chunks
    const uint64_t *chunk_table;
    uint64_t n_chunks;
    // range-filtered batches of gemm_rowreg_kernel: the level's tiles are the ENTRIES of this table (aligned 32-row tile | mask of the
    // wanted rows << 32, build_tile_table_kernel) instead of the tiles themselves; nullptr = every tile, every row
    const uint64_t *tile_table = nullptr;
    // BOOTSTRAP level of gemm_rowreg_kernel (gemm_topk.hip launch_gemm_topk): no nominations -- the kernel stores, per visited tile and
    // query, the best nominating distance of the tile's wanted rows, folded by minimum into tile_min[level tile index % 1024][nqt * 32]
    // (preset to +inf)
    float *tile_min = nullptr;
    const void *image;            // gemm_rowreg_kernel<MODE, true>: the corpus' fp16 operand image (16 KiB per 32-row tile) ...
    const uint32_t *image_zero;   // ... and per tile the mask of its zero rows
    int buffered;                 // gemm_rowreg_kernel: nominations go through the wave's LDS buffer (every level but the first)
    unsigned long long *stamps;   // trace builds only (SMT_RR_EXP & 256, tools/exp_k3_trace.sh): s_memtime stamps of one block
};

__device__ __forceinline__ uint64_t level_tile(uint64_t i, uint64_t stride, int skip16)
{
    if (!skip16) return i * stride;
    // i-th positive integer that is not a multiple of the ratio (constant divisors: no runtime division)
    const uint64_t d = skip16 == 64 ? i / 63 : skip16 == 16 ? i / 15 : i / 3;
    const uint64_t u = d * (uint64_t)skip16 + (i - d * (uint64_t)(skip16 - 1)) + 1;
    return u * stride;
}

__device__ __forceinline__ void append_candidates(const f32x16 &acc, unsigned zero16, unsigned valid16, uint32_t q,
                                                  float thr, float rq, uint64_t row0, int h, key_t64 *cand, unsigned int *counts);
// The candidate test in the SCORE domain.  The row tile is scaled by 1/|row| once when it is loaded, so an
// accumulator is already cos * |q|; "distance <= tau" becomes acc >= (1 - tau) / |q|^-1 ... i.e. ONE compare
// per (row, query) in the epilogue instead of two multiplies, a subtract, a max and a compare (the epilogue
// cost 5 % of a 1000 x 10 M batch).  The bound is lowered by two ulps: a borderline row is admitted rather
// than lost (candidates are nominations; the final distances are exact).  Zero query (rq == 0): every
// distance is 1 (0 against a zero row), the slot then carries tau itself.
__device__ __forceinline__ float score_threshold(float tau, float rq)
{
    if (rq == 0.0f) return tau;
    const float t = (1.0f - tau) / rq;  // tau = +inf (first level) -> -inf: everything passes
    return t - fabsf(t) * 2.4e-7f;
}

// ---- shared epilogue: lane (j, h) owns query q and the 16 rows acc_row(r, h) of the tile at row0
// which of the lane's 16 scores are nominations (bit r <-> accumulator register r); 0 in every lane when the tile has none
__device__ __forceinline__ float nomination_dist(const f32x16 &acc, int r, unsigned zero16, float rq)
{
    if (rq == 0.0f) return (zero16 >> r) & 1u ? 0.0f : 1.0f;  // zero query: 0 against a zero row, else 1 (simsimd rules)
    return fmaxf(1.0f - acc[r] * rq, 0.0f);                    // a zero row has acc == 0 -> 1
}
__device__ __forceinline__ unsigned nomination_mask(const f32x16 &acc, unsigned zero16, unsigned valid16, float thr, float rq)
{
    // almost every (tile, query tile) nominates nothing: one max over the lane's 16 scores (v_max3) and one wave-wide
    // test skip the per-row work (the per-row compares were 1/4 of a bf16 x 3 tile product)
    {
        float mx = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
        for (int r = 3; r + 1 < 16; r += 2) mx = fmaxf(fmaxf(mx, acc[r]), acc[r + 1]);
        mx = fmaxf(mx, acc[15]);
        if (!__builtin_amdgcn_ballot_w64(rq == 0.0f || mx >= thr)) return 0;
    }
    unsigned pass = 0;
    if (rq != 0.0f) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (acc[r] >= thr) pass |= 1u << r;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (nomination_dist(acc, r, zero16, rq) <= thr) pass |= 1u << r;
    }
    return pass & valid16;
}
// acc[r] for a per-lane r: a select chain (a dynamically indexed register array would live in scratch)
__device__ __forceinline__ float acc_select(const f32x16 &acc, int r)
{
    float a = acc[0];
#pragma unroll
    for (int rr = 1; rr < 16; ++rr) a = r == rr ? acc[rr] : a;
    return a;
}
// The nominations of a tile are walked lowest-bit-first in a wave-uniform loop: one trip as a rule (a lane rarely holds two).
// (The obvious form -- sixteen "if (pass & bit)" blocks -- is sixteen exec-masked regions with a taken branch around each:
// measured ~1150 cycles per nominating product at 1000 x 10 M, during which the other seven waves stood at the ring barrier.)
// straight to the per-query lists: one slot grab per lane per tile
__device__ __forceinline__ void append_direct(const f32x16 &acc, unsigned pass, unsigned zero16, uint32_t q, float rq, uint64_t row0, int h,
                                              key_t64 *cand, unsigned int *counts)
{
    unsigned slot = 0;
    if (pass) slot = atomicAdd(&counts[q], (unsigned)__popc(pass));
    key_t64 *dst = cand + (size_t)q * CAND_CAP;
    unsigned todo = pass;
    while (__builtin_amdgcn_ballot_w64(todo != 0)) {
        const int r = todo ? __builtin_ctz(todo) : 0;
        const float a = acc_select(acc, r);
        if (todo) {
            const float d = rq == 0.0f ? ((zero16 >> r) & 1u ? 0.0f : 1.0f) : fmaxf(1.0f - a * rq, 0.0f);
            if (slot < CAND_CAP) dst[slot] = make_key(d, (uint32_t)(row0 + acc_row(r, h)));
            ++slot;
            todo &= todo - 1;
        }
    }
}
__device__ __forceinline__ void append_candidates(const f32x16 &acc, unsigned zero16, unsigned valid16, uint32_t q,
                                                  float thr, float rq, uint64_t row0, int h, key_t64 *cand, unsigned int *counts)
{
    const unsigned pass = nomination_mask(acc, zero16, valid16, thr, rq);
    if (__builtin_amdgcn_ballot_w64(pass != 0)) append_direct(acc, pass, zero16, q, rq, row0, h, cand, counts);
}

// Per query: keep the kp best of the candidates gathered so far (sorted, at the
// head of the buffer), publish tau = kp-th distance, flag overflow.
struct LevelSelectParams {
    key_t64 *cand;
    unsigned int *counts;
    float *tau;
    unsigned int *overflow;  // [nq], sticky
    uint32_t kp;
    float *qconst;           // when set: [q][2] = (score threshold of the new tau, 1/|q|), what gemm_rowreg_kernel reads
};

// ---- gemm_rowreg_kernel (gemm_rowreg.hip): what the host side plans with
constexpr int RR_TROW = 80;                       // bytes per row in the transpose buffer (64 used)
constexpr int RR_TBUF = 32 * RR_TROW;             // per wave
// Streaming (more than four query tiles): the four slots form a ring of single tiles -- tile n is multiplied while
// tiles n+1 .. n+3 are in flight or landed (the DMA of n+3 is issued during product n), one barrier per tile.  A
// distance of one step (the pair scheme of gemm_level_kernel, or two slots per block with two blocks per CU -- both
// measured) leaves the L2 -> LDS latency of every tile exposed at the barrier: a step is only 1.5-3 k cycles of
// bf16 MFMAs, no longer the 16 k of the f32 kernel.
constexpr int RR_THREADS = 512;
constexpr int RR_WAVES = RR_THREADS / 64;
constexpr int RR_SLOTS = 4;
#ifndef SMT_RR_BDIST
#define SMT_RR_BDIST 2
#endif
constexpr int RR_BDIST = SMT_RR_BDIST;            // K-steps between the LDS read of a B quad pair and its MFMAs
constexpr int RR_QCONST = QT_ROWS * 8;            // per slot: (score threshold, 1/|q|) of the tile's 32 queries
// NOMINATIONS GO THROUGH LDS.  During a sweep the wave's transpose buffer is idle; it holds the nominations of the sweep --
// RR_CB_CAP (key, query) pairs, allocated with ballot / readlane arithmetic (no atomic, no memory wait) -- and the wave flushes
// them to the per-query lists (one returning atomic per pair, then a store) at the START OF THE NEXT ROW PHASE, behind the 32 row
// loads it has to wait for anyway.  Before (wave timeline, tools/trace_k3.py, 1000 x 10 M): the main level admits ~16 (k + 24)
// rows per query (its thresholds come from a 1/16 sample), 6 % of the products nominate something, and the direct path --
// global_atomic_add with return, s_waitcnt vmcnt(0), then a vmcnt(0) in front of every store; vmcnt is in-order, so each of these
// also waits for the query-tile DMAs in flight -- held its wave for ~2000 cycles while the other seven waited at the ring's
// barrier: 63 % of the barriers had such a straggler, the barrier period was 5200 cycles instead of 4350.
constexpr int RR_CB_CAP = 208;                    // 208 x 8 B keys + 208 x 4 B queries = 2496 B <= RR_TBUF
// Per nomination mode: how a query tile lies in LDS and how deep the ring is.  bf16 x 3 / f16 x 2 read a hi and a lo
// quad per (K-step, half): 1 KiB per query (65-float4 rows), four slots.  f16 x 1 reads the hi quads only: its image is
// COMPACT -- 512 B per query, rows UNPADDED and swizzled (chunk c of row j at c ^ (j & 15): conflict-free b128 reads, and two rows
// are contiguous, so one full-wave LDS-DMA instruction stages a pair: gemm_rowreg.hip STAGE_N) -- so EIGHT slots fit
// the same LDS: batches of up to 256 queries stay resident (no ring, no barrier), and a streamed batch has seven tiles
// in flight or landed instead of three (a ring step is 16 MFMAs per wave now, half of f16 x 2's: three steps no longer
// cover the L2 -> LDS latency of a tile).
#ifndef SMT_RR_GT
#define SMT_RR_GT 2
#endif
template <int MODE>
struct RrGeom {
    static constexpr int SLOTS = MODE == 2 ? 8 : RR_SLOTS;
    static constexpr int ROW_F4 = MODE == 2 ? 32 : QT_STRIDE_F4;     // float4 per query row in LDS (f16 x 1: unpadded, swizzled)
    static constexpr int SLOT_F4 = QT_ROWS * ROW_F4;                  // float4 per slot
    static constexpr int QUERY_WORDS = MODE == 2 ? 128 : 256;         // words per query in the global split image
    // The ring advances in GROUPS of GT tiles: one block-wide barrier per group instead of per tile (between barriers the
    // eight waves run free -- the barrier is what kept them in lock step, row phases included).  Tile pos + AHEAD is staged
    // during product pos into the slot that tile pos + AHEAD - SLOTS used: that one must belong to an EARLIER group than
    // pos (every wave is past it), hence AHEAD = SLOTS - GT; at a group border the tiles of the next group were staged at
    // least AHEAD - GT + 1 products ago.
    static constexpr int GT = MODE == 2 ? SMT_RR_GT : 1;
    static constexpr int AHEAD = SLOTS - GT;
    static constexpr int SMEM = SLOTS * SLOT_F4 * 16 + RR_WAVES * RR_TBUF + SLOTS * RR_QCONST;
};
constexpr int RR_SMEM = RrGeom<0>::SMEM;

// ---- gemm_ldsrow_kernel (gemm_ldsrow.hip)
constexpr int LR_THREADS = 512;
constexpr int LR_WAVES = LR_THREADS / 64;
constexpr int LR_SLICE_BYTES = 32 * 256;           // one K-slice: 32 rows x 64 dims = 8 KiB
constexpr int LR_QTILE_BYTES = 32 * 1024;          // NQT = 2: a query tile in LDS, swizzled, unpadded

template <int NQT>
struct LrGeom {
    static constexpr int SLOTS = NQT == 1 ? 2 : 1;                        // ring slots per wave
    static constexpr int Q_BYTES = NQT == 1 ? 0 : NQT * LR_QTILE_BYTES;   // queries in LDS (NQT = 1: in registers)
    static constexpr int RING_BYTES = SLOTS * LR_SLICE_BYTES;             // per wave
    static constexpr int SCRATCH_OFF = Q_BYTES + LR_WAVES * RING_BYTES;   // 128 B per wave: row scales for the epilogue
    static constexpr int SMEM = SCRATCH_OFF + LR_WAVES * 128;
};

// ---- launchers of the kernel families (no profiling brackets, no error check: the caller owns both)
hipError_t gemm_level_set_attrs();
size_t gemm_level_smem_bytes(uint32_t nqt);
void gemm_level_launch(smt_ctx *ctx, bool bf16, uint32_t nqt, int nb, const GemmParams &g);
hipError_t gemm_rowreg_set_attrs();
void gemm_rowreg_launch(smt_ctx *ctx, int mode /* 0 bf16 x 3, 1 f16 x 2, 2 f16 x 1 */, bool image, int nb, const GemmParams &g);
hipError_t gemm_ldsrow_set_attrs();
void gemm_ldsrow_launch(smt_ctx *ctx, bool bf16, uint32_t nqt, bool filtered, bool nt, int nb, const GemmParams &g);

}  // namespace smt
