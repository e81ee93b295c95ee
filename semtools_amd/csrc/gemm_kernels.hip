// gemm_kernels.hip -- K3: batched queries.  S = C x Q^T on the MFMA pipes with the top-k candidate selection fused
// into the epilogue -- the nq x N score matrix (40 GB at 1k x 10M) is never materialised.  The scores only NOMINATE
// candidates (k + guard per query); answers are re-scored exactly in f64 and proved complete by the select stage.
//
// Kernels in this file, and who runs when (launch_gemm_topk):
//   gemm_rowreg_kernel<F16X2>   DEFAULT for every unfiltered batch.  Row tiles arrive with coalesced loads, are split
//                               into 16-bit operands once and transposed through LDS; bf16 x 3 products below 128
//                               queries, f16 x 2 from there (tuning key gemm_nominate).  DESIGN.md 4.3c.
//   gemm_ldsrow_kernel<..>      range-filtered batches (chunk table), <= 64 queries per pass; f32 or bf16 x 3 MFMAs.
//   gemm_level_kernel<BF16>     the round-1 corpus-stationary kernel: f32 MFMAs (v_mfma_f32_32x32x2_f32, exact f32,
//                               157 TF peak) when gemm_bf16x3 = 0, or bf16 x 3 when gemm_rowreg = 0.  The level scheme,
//                               the epilogue and the candidate buffers described below are shared by all three.
//   level_select_kernel, split_queries_*_kernel, query_consts_kernel: per-level / per-batch helpers.
//   launch_gemm_threshold       one sweep with preset thresholds: the batched exhaustive re-answer (api.cpp).
//
// The f32 design (gemm_level_kernel):
//
// No reference counterpart: the reference answers one query per process with a
// scalar loop (src/search/mod.rs:84-86).  Contract = same results as the K2
// scan / the oracle for every query of the batch.
//
// Decomposition (corpus-stationary):
//   * a wave owns one ROW TILE = 32 corpus rows and keeps it in REGISTERS for the
//     whole sweep over the queries (128 VGPRs: 32 rows x 256 dims / 64 lanes);
//   * the queries stream through LDS in QUERY TILES of 32 (32.5 KiB each): four slots hold the PAIR of tiles
//     being multiplied and the pair being fetched (LDS-DMA loads), one barrier per pair, shared by the
//     block's 8 waves; resident when a block sweeps <= 4 tiles;
//   * per (row tile, query tile): 128 MFMAs (K = 256 in steps of 2) accumulate
//     a 32x32 block in 16 accumulator VGPRs.  A = corpus rows, B = queries, so
//     each LANE owns one query (column) and 16 rows: the candidate test is
//     lane-local -- the row tile was scaled by 1/|c_row| when it was loaded, so
//     the test is one compare of the accumulator with the query's score bound
//     (1 - tau) * |q|; on a (rare) hit (distance,row) goes to the query's
//     candidate buffer with one atomic slot grab.
//   * K order: lane l < 32 feeds dims 8m..8m+3 and lane l >= 32 dims 8m+4..8m+7 of
//     instruction group m (one 16-B load per 4 MFMAs for either operand).  The
//     same permutation is applied to A and B, so the dot product is unchanged.
//
// Thresholds: the row tiles are visited in LEVELS (every 16^j-th tile first).
// Level 0 is small and appends everything; after each level a select kernel
// keeps each query's best kp candidates and sets tau = its kp-th distance,
// which upper-bounds the final kp-th distance, so later levels append only
// about 16*kp candidates per query.  Every tile is processed exactly once.
// The candidate SET depends only on the data (never on timing), and the final
// answer is the exact top-k of a superset of the true top-k' => deterministic.
// A query whose buffer overflows (adversarial row order) is flagged and redone
// by the K2 scan.
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
constexpr int LEVEL_RATIO_SMALL_K = 16;
constexpr int LEVEL_RATIO_LARGE_K = 16;
constexpr uint32_t LEVEL_RATIO_KP_LIMIT = 24;
constexpr int LEVEL0_MAX_TILES = 32;
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
    int skip16;               // LEVEL_RATIO (64 or 16) when u skips the multiples of the ratio (they belong to earlier levels), else 0
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
    const void *image;            // gemm_rowreg_kernel<MODE, true>: the corpus' fp16 operand image (16 KiB per 32-row tile) ...
    const uint32_t *image_zero;   // ... and per tile the mask of its zero rows
    int buffered;                 // gemm_rowreg_kernel: nominations go through the wave's LDS buffer (every level but the first)
    unsigned long long *stamps;   // trace builds only (SMT_RR_EXP & 256, tools/exp_k3_trace.sh): s_memtime stamps of one block
};

__device__ __forceinline__ uint64_t level_tile(uint64_t i, uint64_t stride, int skip16)
{
    if (!skip16) return i * stride;
    // i-th positive integer that is not a multiple of the ratio (constant divisors: no runtime division)
    const uint64_t d = skip16 == 64 ? i / 63 : i / 15;
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

// Split image of the queries for the bf16 x 3 kernels: one 1 KiB row per query (zero rows pad the last tile);
// K-step m, half h occupy bytes (2m + h) * 32 ..: 16 B of hi (dims 16m + 8h .. + 7 as bf16 pairs) then 16 B of lo --
// exactly the two B-operand quads lane (j, h) feeds to K-step m, so a staged row is read with two ds_read_b128.
__global__ void split_queries_kernel(const float *queries, uint32_t nq, uint32_t nq_pad, uint32_t *out)
{
    const uint32_t q = blockIdx.x * 2 + (threadIdx.x >> 7), pr = threadIdx.x & 127;  // pair pr = dims 2pr, 2pr + 1
    if (q >= nq_pad) return;
    uint32_t hi = 0, lo = 0;
    if (q < nq) {
        const f32x2 v = reinterpret_cast<const f32x2 *>(queries + (size_t)q * 256)[pr];
        bf16_split2(v.x, v.y, hi, lo);
    }
    uint32_t *row = out + (size_t)q * 256 + (pr >> 2) * 8 + (pr & 3);
    row[0] = hi;
    row[4] = lo;
}

template <bool BF16>
__global__ void __launch_bounds__(GEMM_THREADS, 2) gemm_level_kernel(GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4 *s_q = reinterpret_cast<f32x4 *>(smem_raw);                  // [4][32][65] float4: two PAIRS of query tiles
    float *s_tau = reinterpret_cast<float *>(s_q + 4 * QT_F4);        // [nqt*32]
    float *s_rq = s_tau + (size_t)p.nqt * QT_ROWS;                    // [nqt*32]  1/|q| (0 for a zero query)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    // small levels have fewer row-tile groups than CUs: qsplit blocks share one group and split the query tiles
    const uint32_t qs = blockIdx.x % p.qsplit;
    const uint32_t row_block = blockIdx.x / p.qsplit, row_blocks = gridDim.x / p.qsplit;
    const uint32_t qt_lo = (uint32_t)((uint64_t)qs * p.nqt / p.qsplit);
    const uint32_t qt_hi = (uint32_t)((uint64_t)(qs + 1) * p.nqt / p.qsplit);
    const uint32_t q_lo = qt_lo * QT_ROWS, q_hi = qt_hi * QT_ROWS;
    const uint32_t n_qt = qt_hi - qt_lo;
    const bool resident = n_qt <= 4;  // all of this block's query tiles live in LDS for the whole kernel

    // ---- per-query constants: tau and 1/|q|
    for (uint32_t q = q_lo + wave; q < q_hi; q += GEMM_WAVES) {
        float rq = 0.0f;
        if (q < p.nq) {
            const f32x4 v = reinterpret_cast<const f32x4 *>(p.queries + (size_t)q * 256)[lane];
            const float a2 = wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
            rq = a2 == 0.0f ? 0.0f : __frsqrt_rn(a2);
        }
        if (lane == 0) {
            s_rq[q] = rq;
            s_tau[q] = score_threshold(q < p.nq ? p.tau[q] : -1.0f, rq);  // padding: zero query, tau < 0 -> never passes
        }
    }

    // ---- stage query tile qt into LDS slot `slot` with LDS-DMA loads (global_load_lds_dwordx4: one wave
    // instruction moves one 1 KiB query row straight into its 1040-B LDS row -- no staging registers, no ds_write;
    // the old path went global -> 4 VGPR quads -> 4 ds_write_b128 per thread and cost 1.5 ms of a 43 ms batch).
    // Each wave owns 4 of the tile's 32 rows.  Completion is tracked by vmcnt: wait before the barrier.
    auto stage_tile = [&](uint32_t qt, int slot) {
#pragma unroll
        for (int u = 0; u < QT_ROWS / GEMM_WAVES; ++u) {
            const int r = wave * (QT_ROWS / GEMM_WAVES) + u;  // wave-uniform
            const uint32_t q = qt * QT_ROWS + r;
            f32x4 *dst = s_q + slot * QT_F4 + r * QT_STRIDE_F4;
            if (q < p.nq) {
                const float *src = BF16 ? reinterpret_cast<const float *>(p.queries_split) : p.queries;  // same row size
                __builtin_amdgcn_global_load_lds(src + (size_t)q * 256 + lane * 4,
                                                 (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            } else {
                dst[lane] = (f32x4){0.f, 0.f, 0.f, 0.f};  // padding rows of the last tile
            }
        }
    };
    auto stage_wait = [&]() { __builtin_amdgcn_s_waitcnt(0x0F70); };  // vmcnt(0): the DMA writes have landed
    {
        // resident: tiles 0..n_qt-1 -> slots 0..n_qt-1; streaming: the first pair -> slots 0, 1
        const uint32_t first = resident ? n_qt : 2u;
        for (uint32_t t = 0; t < first; ++t) stage_tile(qt_lo + t, (int)t);
        stage_wait();
    }
    __syncthreads();

    const uint64_t W = (uint64_t)row_blocks * GEMM_WAVES;
    const uint64_t steps = (p.level_tiles + W - 1) / W;  // block-uniform trip count
    uint64_t it = (uint64_t)row_block * GEMM_WAVES + wave;
    int cur = 0;  // LDS buffer holding the current query tile (streaming mode)

    for (uint64_t step = 0; step < steps; ++step, it += W) {
        const bool has = it < p.level_tiles;  // wave-uniform
        const uint64_t row0 = (has ? level_tile(it, p.stride, p.skip16) : 0) * 32;

        // ---- A operand: this wave's 32 corpus rows, register resident (f32: 128 VGPRs; bf16 x 3: 64 hi + 64 lo)
        f32x4 A[BF16 ? 1 : 32];
        u32x4 Ah[BF16 ? 16 : 1], Al[BF16 ? 16 : 1];
        unsigned zero16 = 0;   // bit r: tile row acc_row(r, h) is the zero vector
        unsigned valid16 = 0;  // bit r: that row exists
        if (has) {
            const uint64_t my_row = row0 + j;
            const bool row_ok = my_row < p.n_rows;
            float part = 0.0f;
            float rb;
            if constexpr (!BF16) {
                const f32x4 *src = reinterpret_cast<const f32x4 *>(p.corpus + (row_ok ? my_row : 0) * 256) + h;
#pragma unroll
                for (int m = 0; m < 32; ++m) {
                    A[m] = __builtin_nontemporal_load(src + 2 * m);  // unconditional (address clamped above): 32 loads in flight
                }
                if (!row_ok) {  // rows past the end of the corpus contribute zeros (one test, not one branch per load)
#pragma unroll
                    for (int m = 0; m < 32; ++m) A[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int m = 0; m < 32; ++m)
                    part += A[m].x * A[m].x + A[m].y * A[m].y + A[m].z * A[m].z + A[m].w * A[m].w;
                const float b2 = part + __shfl_xor(part, 32);
                rb = b2 == 0.0f ? 0.0f : __frsqrt_rn(b2);  // row l&31, same in both halves
#pragma unroll
                for (int m = 0; m < 32; ++m) A[m] *= rb;  // unit rows: the accumulators are cosines times |q|
            } else {
                // lane (j, h): dims 16m + 8h .. + 7 of K-step m = float4 4m + 2h and the next one (32 contiguous bytes)
                const f32x4 *src = reinterpret_cast<const f32x4 *>(p.corpus + (row_ok ? my_row : 0) * 256) + 2 * h;
                f32x4 R[32];
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    R[2 * m] = __builtin_nontemporal_load(src + 4 * m);
                    R[2 * m + 1] = __builtin_nontemporal_load(src + 4 * m + 1);
                }
                if (!row_ok) {
#pragma unroll
                    for (int m = 0; m < 32; ++m) R[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int m = 0; m < 32; ++m)
                    part += R[m].x * R[m].x + R[m].y * R[m].y + R[m].z * R[m].z + R[m].w * R[m].w;
                const float b2 = part + __shfl_xor(part, 32);
                rb = b2 == 0.0f ? 0.0f : __frsqrt_rn(b2);
#pragma unroll
                for (int m = 0; m < 16; ++m) bf16_split8(R[2 * m] * rb, R[2 * m + 1] * rb, Ah[m], Al[m]);  // unit rows, split once per tile
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * h;  // accumulator reg r <-> tile row i
                if (__shfl(rb, i) == 0.0f) zero16 |= 1u << r;
                if (row0 + i < p.n_rows) valid16 |= 1u << r;
            }
        }

        // f32 MFMA: one (row tile x query tile) product + epilogue; the tile sits in LDS slot `slot`
        auto tile_product = [&](uint32_t qt, int slot) __attribute__((always_inline)) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            // lane owns query q = qt*32 + j: its threshold and 1/|q| are fetched now, under the MFMAs
            const uint32_t q = qt * QT_ROWS + j;
            const float thr_q = s_tau[q], rq_q = s_rq[q];
            if constexpr (!BF16) {
                const f32x4 *bq = s_q + slot * QT_F4 + j * QT_STRIDE_F4 + h;
#pragma unroll
                for (int m = 0; m < 32; ++m) {
                    const f32x4 b = bq[2 * m];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].x, b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].y, b.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].z, b.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].w, b.w, acc, 0, 0, 0);
                }
            }
            append_candidates(acc, zero16, valid16, q, thr_q, rq_q, row0, h, p.cand, p.counts);  // 16 rows x this lane's query
        };
        // bf16 x 3: NT (1 or 2) query tiles against the row tile at once.  The B quads are software-pipelined through
        // registers (LDS latency is ~100 cycles, a K-step of one tile is only 3 MFMAs = 96): prefetch distance 2 K-steps
        // for one tile, 1 for two; two tiles also interleave their accumulators, so no MFMA waits for its predecessor.
        auto tile_products_bf16 = [&](auto NTc, uint32_t qt0, int slot0) __attribute__((always_inline)) {
            constexpr int NT = decltype(NTc)::value;
            constexpr int D = NT == 1 ? 2 : 1, NB = D + 1;
            f32x16 acc[NT];
            const u32x4 *bq[NT];
            float thr_q[NT], rq_q[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
                bq[t] = reinterpret_cast<const u32x4 *>(s_q + (slot0 + t) * QT_F4 + j * QT_STRIDE_F4) + 2 * h;
            }
            u32x4 bh[NB][NT], bl[NB][NT];
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int t = 0; t < NT; ++t) { bh[d][t] = bq[t][4 * d]; bl[d][t] = bq[t][4 * d + 1]; }
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (m + D < 16) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) { bh[(m + D) % NB][t] = bq[t][4 * (m + D)]; bl[(m + D) % NB][t] = bq[t][4 * (m + D) + 1]; }
                } else if (m + D == 16) {
                    // the lane's query constants arrive under the last MFMAs
#pragma unroll
                    for (int t = 0; t < NT; ++t) { thr_q[t] = s_tau[(qt0 + t) * QT_ROWS + j]; rq_q[t] = s_rq[(qt0 + t) * QT_ROWS + j]; }
                }
                const bf16x8 ah = __builtin_bit_cast(bf16x8, Ah[m]), al = __builtin_bit_cast(bf16x8, Al[m]);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, bh[m % NB][t]), acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, __builtin_bit_cast(bf16x8, bh[m % NB][t]), acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, bl[m % NB][t]), acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);  // keeps the prefetch distance: hipcc otherwise sinks each read to its use
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
                append_candidates(acc[t], zero16, valid16, (qt0 + t) * QT_ROWS + j, thr_q[t], rq_q[t], row0, h, p.cand, p.counts);
        };
        using One = std::integral_constant<int, 1>;
        // tiles [t, t + n) of this block's range sit in consecutive slots from `slot`
        auto products = [&](uint32_t t, uint32_t n, int slot) __attribute__((always_inline)) {
            if constexpr (BF16) {
                // (two tiles at once with interleaved accumulators measured no faster at 1000 queries and slower at
                // 128 -- the second accumulator and B buffers push the kernel into spills)
                tile_products_bf16(One{}, qt_lo + t, slot);
                if (n == 2) tile_products_bf16(One{}, qt_lo + t + 1, slot + 1);
            } else {
                tile_product(qt_lo + t, slot);
                if (n == 2) tile_product(qt_lo + t + 1, slot + 1);
            }
        };
        if (resident) {
            if (has)
                for (uint32_t t = 0; t < n_qt; t += 2) products(t, t + 1 < n_qt ? 2u : 1u, (int)t);
        } else {
            // streaming: TWO query tiles per barrier (the block's 8 waves meet half as often: the barrier cost
            // 2.6 ms of a 43 ms batch).  While pair P is multiplied, the next pair lands in the other two slots.
            for (uint32_t t = 0; t < n_qt; t += 2) {
                const bool two = t + 1 < n_qt;  // block-uniform
                const uint32_t tn0 = (t + 2) % n_qt, tn1 = (t + 3) % n_qt;  // wraps into the next row tile's sweep
                // (when n_qt is odd the last pair holds one tile: the next sweep restarts at tile 0 in slot 0)
                const uint32_t nx0 = two ? tn0 : 0u, nx1 = two ? tn1 : 1u;
                stage_tile(qt_lo + nx0, (cur ^ 1) * 2);      // both DMA batches fly under the two products
                stage_tile(qt_lo + nx1, (cur ^ 1) * 2 + 1);
                if (has) products(t, two ? 2u : 1u, cur * 2);
                stage_wait();
                __syncthreads();
                cur ^= 1;
            }
        }
    }
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

// ---- bf16 x 3, every unfiltered batch size: row tiles arrive in REGISTERS with coalesced loads and are transposed
// into the MFMA operand layout through a small wave-private LDS buffer.
//
// Why: the MFMA A layout wants lane j <-> row j, and loading a row tile directly in that layout means 32 instructions
// that each touch 16- or 32-byte pieces of 32 different rows.  Measured on MI355X (tools/micro/row_load_patterns.hip,
// 8 waves per CU, 10 M rows): such loads deliver 3.2-3.6 TB/s whatever the sweep behind them; instructions that read
// 128-byte runs (8 lanes per row, 8 rows per instruction) deliver 7.0-7.2 TB/s, also with 192 MFMAs per tile behind
// them.  So: 32 coalesced loads per tile -> 1/|row| scaling and the bf16 hi/lo split ONCE per tile in that layout ->
// per 32-dim slice the packed words go through a 2.5 KiB LDS buffer (80-byte row stride: conflict-free b128 reads)
// and come back as the operand quads Ah/Al[16], which then serve the whole sweep over the query tiles.  The query
// tiles are the split image (split_queries_kernel) in LDS: resident up to 4 tiles, else streamed in pairs like
// gemm_level_kernel.  Thresholds and 1/|q| are read from global memory (kept per level by level_select_kernel), so
// the LDS footprint does not depend on the batch size.
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
// COMPACT -- 512 B per query, 33-float4 rows (132 words: the same 4-bank step per lane as 260) -- so EIGHT slots fit
// the same LDS: batches of up to 256 queries stay resident (no ring, no barrier), and a streamed batch has seven tiles
// in flight or landed instead of three (a ring step is 16 MFMAs per wave now, half of f16 x 2's: three steps no longer
// cover the L2 -> LDS latency of a tile).
#ifndef SMT_RR_GT
#define SMT_RR_GT 2
#endif
template <int MODE>
struct RrGeom {
    static constexpr int SLOTS = MODE == 2 ? 8 : RR_SLOTS;
    static constexpr int ROW_F4 = MODE == 2 ? 33 : QT_STRIDE_F4;     // float4 per query row in LDS
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

// The f16 x 2 image of the queries (same row layout: K-step m, half h -> 16 B of hi, 16 B of lo): the UNIT query times
// 2^8, split into two fp16 parts.  One wave per query (the norm is needed first).
__global__ void split_queries_f16_kernel(const float *queries, uint32_t nq, uint32_t nq_pad, uint32_t *out)
{
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= nq_pad) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < nq) v = reinterpret_cast<const f32x4 *>(queries + (size_t)q * 256)[lane];
    const float a2 = wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
    const float s = a2 == 0.0f ? 0.0f : __frsqrt_rn(a2) * F16X2_QUERY_SCALE;
    uint32_t h0, l0, h1, l1;
    f16_split2(v.x * s, v.y * s, h0, l0);   // pairs 2 lane, 2 lane + 1  (dims 4 lane .. 4 lane + 3)
    f16_split2(v.z * s, v.w * s, h1, l1);
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    uint32_t *row = out + (size_t)q * 256 + (lane >> 1) * 8 + 2 * (lane & 1);   // pair pr -> word (pr >> 2) * 8 + (pr & 3)
    *reinterpret_cast<u32x2 *>(row) = (u32x2){h0, h1};
    *reinterpret_cast<u32x2 *>(row + 4) = (u32x2){l0, l1};
}

// The f16 x 1 image: the hi halves only, 512 B per query -- (K-step m, half h) -> 16 B at word 8 m + 4 h.
__global__ void split_queries_f16x1_kernel(const float *queries, uint32_t nq, uint32_t nq_pad, uint32_t *out)
{
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= nq_pad) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < nq) v = reinterpret_cast<const f32x4 *>(queries + (size_t)q * 256)[lane];
    const float a2 = wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
    const float s = a2 == 0.0f ? 0.0f : __frsqrt_rn(a2) * F16X2_QUERY_SCALE;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    // lane holds dims 4 lane .. 4 lane + 3 = pairs 2 lane, 2 lane + 1 -> words 2 lane, 2 lane + 1 of the 128
    *reinterpret_cast<u32x2 *>(out + (size_t)q * 128 + 2 * lane) = (u32x2){f16_pack2(v.x * s, v.y * s), f16_pack2(v.z * s, v.w * s)};
}

__global__ void query_consts_kernel(const float *queries, uint32_t nq, uint32_t nq_pad, float *qconst, int f16x2)
{
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= nq_pad) return;
    float rq = 0.0f;
    if (q < nq) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(queries + (size_t)q * 256)[lane];
        const float a2 = wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
        rq = a2 == 0.0f ? 0.0f : __frsqrt_rn(a2);
    }
    if (f16x2 && rq != 0.0f) rq = F16X2_INV_SCALE;   // the f16 x 2 operands are unit vectors times 2^10 and 2^8
    if (lane == 0) {
        qconst[2 * q] = score_threshold(q < nq ? __builtin_inff() : -1.0f, rq);  // padding: zero query, tau < 0 -> never passes
        qconst[2 * q + 1] = rq;
    }
}

// MODE 0: bf16 x 3.  MODE 1: f16 x 2 nomination (mfma_tile.h) -- the rows carry ONE fp16 operand (64 VGPRs), the transpose
// moves half the words, a K-step is two MFMAs (row x query-hi, row x query-lo).  MODE 2: f16 x 1 -- the query's lo part is
// dropped too: ONE MFMA and ONE B quad per K-step, half the MFMA and half the LDS operand traffic of f16 x 2 for a
// certificate band of 2^-10 instead of 2^-11 (common.h F32_ERR_F16X1): the large-batch mode, where the MFMA pipe -- at the
// clock the part sustains under this load -- is the bound and the only lever left is fewer MFMAs per useful flop.
#if defined(SMT_RR_EXP) && (SMT_RR_EXP & 256)
#define CB_STAMP(id) do { if (dbg) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) dbg[dbg_n] = (t_ << 8) | (unsigned)(id); ++dbg_n; } } while (0)
#define CB_DBG_PARAMS , unsigned long long *dbg, int &dbg_n
#else
#define CB_STAMP(id) do { } while (0)
#define CB_DBG_PARAMS
#endif
// the wave's nomination buffer (its transpose buffer, idle during a sweep): keys at cb, queries behind them; n_buf is wave-uniform
__device__ __forceinline__ void append_candidates_lds(const f32x16 &acc, unsigned zero16, unsigned valid16, uint32_t q, float thr, float rq,
                                                      uint64_t row0, int h, int lane, unsigned char *cb, uint32_t &n_buf,
                                                      key_t64 *cand, unsigned int *counts CB_DBG_PARAMS)
{
    // Every VALU instruction of an epilogue competes with the MFMA stream of the SIMD's other wave (~12 cycles apiece there), and
    // while it runs the block's other waves may be standing at the ring barrier.  So, in order of frequency:
    //  (1) nothing to nominate (~90 % of the products at 1000 x 10 M): eight v_max3, one compare, one branch;
    //  (2) nominations of nonzero queries in a full tile: one v_cmp per accumulator register, its lane mask in SGPRs -- scalar
    //      tests skip the registers without a nomination; a register with some: slot = count + mbcnt, two LDS writes;
    //  (3) zero queries with a reachable threshold, tiles that hang over the end of the corpus: the general mask, bit by bit.
    const bool zq = rq == 0.0f;    // a zero query's slot carries tau itself (score_threshold): padding has tau < 0
    {
        float mx = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
        for (int r = 3; r + 1 < 16; r += 2) mx = fmaxf(fmaxf(mx, acc[r]), acc[r + 1]);
        mx = fmaxf(mx, acc[15]);
        if (!__builtin_amdgcn_ballot_w64(zq ? thr >= 0.0f : mx >= thr)) return;
    }
    CB_STAMP(11);
    key_t64 *keys = reinterpret_cast<key_t64 *>(cb);
    uint32_t *qs = reinterpret_cast<uint32_t *>(cb + RR_CB_CAP * 8);
    if (__builtin_amdgcn_ballot_w64((zq && thr >= 0.0f) || valid16 != 0xffffu)) {
        unsigned pass = 0;
        if (!zq) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (acc[r] >= thr) pass |= 1u << r;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (nomination_dist(acc, r, zero16, rq) <= thr) pass |= 1u << r;
        }
        pass &= valid16;
        const unsigned long long lanes = __builtin_amdgcn_ballot_w64(pass != 0);
        if (!lanes) return;
        const uint32_t mine = (uint32_t)__popc(pass);
        uint32_t slot = 0, run = n_buf;
        for (unsigned long long m = lanes; m; m &= m - 1) {
            const int l = __builtin_ctzll(m);
            if (lane == l) slot = run;
            run += (uint32_t)__builtin_amdgcn_readlane((int)mine, l);
        }
        if (run > (uint32_t)RR_CB_CAP) {   // wave-uniform: no room: this tile goes straight to the lists
            append_direct(acc, pass, zero16, q, rq, row0, h, cand, counts);
            return;
        }
        n_buf = run;
        unsigned todo = pass;
        while (__builtin_amdgcn_ballot_w64(todo != 0)) {
            const int r = todo ? __builtin_ctz(todo) : 0;
            const float a = acc_select(acc, r);
            if (todo) {
                const float d = zq ? ((zero16 >> r) & 1u ? 0.0f : 1.0f) : fmaxf(1.0f - a * rq, 0.0f);
                keys[slot] = make_key(d, (uint32_t)(row0 + acc_row(r, h)));
                qs[slot] = q;
                ++slot;
                todo &= todo - 1;
            }
        }
        return;
    }
    const unsigned long long nonzero_q = __builtin_amdgcn_ballot_w64(!zq);
    const uint32_t lo = (uint32_t)lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned long long mr = __builtin_amdgcn_ballot_w64(acc[r] >= thr) & nonzero_q;
        if (__builtin_expect(mr != 0, 0)) {
            const uint32_t n = (uint32_t)__builtin_popcountll(mr);
            const bool mine = (mr >> lo) & 1ull;
            const key_t64 key = make_key(fmaxf(1.0f - acc[r] * rq, 0.0f), (uint32_t)(row0 + acc_row(r, h)));
            if (n_buf + n <= (uint32_t)RR_CB_CAP) {
                if (mine) {
                    const uint32_t slot = n_buf + __builtin_amdgcn_mbcnt_hi((uint32_t)(mr >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mr, 0u));
                    keys[slot] = key;
                    qs[slot] = q;
                }
                n_buf += n;
            } else if (mine) {   // no room (a loose threshold): straight to the list
                const unsigned slot = atomicAdd(&counts[q], 1u);
                if (slot < CAND_CAP) cand[(size_t)q * CAND_CAP + slot] = key;
            }
        }
    }
    CB_STAMP(13);
}
__device__ __forceinline__ void flush_candidates_lds(int lane, const unsigned char *cb, uint32_t &n_buf, key_t64 *cand, unsigned int *counts)
{
    const key_t64 *keys = reinterpret_cast<const key_t64 *>(cb);
    const uint32_t *qs = reinterpret_cast<const uint32_t *>(cb + RR_CB_CAP * 8);
    for (uint32_t i = (uint32_t)lane; i < n_buf; i += 64) {
        const uint32_t q = qs[i];
        const unsigned slot = atomicAdd(&counts[q], 1u);
        if (slot < CAND_CAP) cand[(size_t)q * CAND_CAP + slot] = keys[i];
    }
    n_buf = 0;
}

// THE CORPUS' fp16 OPERAND IMAGE (smt_corpus::image, api.cpp): per 32-row tile the sixteen operand quads gemm_rowreg_kernel's fp16
// modes build in their row phase -- unit row x 2^10, fp16, quad (K-step m, lane l = 32 h + j) at 16 (64 m + l) bytes of the
// tile's 16 KiB -- written ONCE per row by this kernel with the same arithmetic in the same order (the tests compare the
// nominations of both forms bit for bit), plus the tile's zero-row mask.  One wave per tile.
__global__ void __launch_bounds__(256) pack_image_kernel(const float *corpus, uint64_t n_rows, uint64_t first_tile, uint64_t n_tiles,
                                                         uint32_t *image, uint32_t *image_zero)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_t[4 * RR_TBUF];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const uint64_t t = (uint64_t)blockIdx.x * 4 + wave;
    if (t >= n_tiles) return;
    const uint64_t tile = first_tile + t, row0 = tile * 32;
    unsigned char *tbuf = s_t + wave * RR_TBUF;
    const uint32_t t_wr = (uint32_t)((lane >> 3) * RR_TROW + (lane & 7) * 8);
    const uint32_t t_rd = (uint32_t)(j * RR_TROW + h * 16);
    f32x4 R[32];
    {
        const f32x4 *base = reinterpret_cast<const f32x4 *>(corpus) + (lane & 7);
        uint64_t rowv[4];
        bool inside[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t r = row0 + 8 * u + (lane >> 3);
            inside[u] = r < n_rows;
            rowv[u] = inside[u] ? r : 0;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            R[i] = __builtin_nontemporal_load(base + rowv[i & 3] * 64 + 8 * (i >> 2));
            if (!inside[i & 3]) R[i] = (f32x4){0.f, 0.f, 0.f, 0.f};   // rows past the end: zero rows (masked by valid16 in the sweep)
        }
    }
    float rb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float part = 0.0f;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            const f32x4 v = R[4 * sl + u];
            part += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 4);
        rb[u] = part == 0.0f ? 0.0f : __frsqrt_rn(part);
    }
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x4 *out = reinterpret_cast<u32x4 *>(image) + tile * 1024 + lane;
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 v = R[4 * sl + u] * (rb[u] * F16X2_ROW_SCALE);
            *reinterpret_cast<u32x2 *>(tbuf + t_wr + u * 8 * RR_TROW) = (u32x2){f16_pack2(v.x, v.y), f16_pack2(v.z, v.w)};
        }
        out[(2 * sl) * 64] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd);
        out[(2 * sl + 1) * 64] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd + 32);
    }
    uint32_t zm = 0;   // bit 8u + i: tile row 8u + i (its 1/|row| sits in lanes 8i .. 8i + 7 of rb[u])
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned long long z = __ballot(rb[u] == 0.0f);
#pragma unroll
        for (int i = 0; i < 8; ++i) zm |= (uint32_t)((z >> (8 * i)) & 1ull) << (8 * u + i);
    }
    if (lane == 0) image_zero[tile] = zm;
}

#ifndef SMT_RR_EXP
#define SMT_RR_EXP 0
#endif
#if (SMT_RR_EXP & 256)
// wave timeline of block 40, second step, main level (level_tiles large): (s_memtime << 8) | id, 1024 stamps per wave
#define RR_STAMP(id) do { if (tracing && n_stamp < 1024) { const unsigned long long t_ = __builtin_readcyclecounter(); \
        if (lane == 0) p.stamps[wave * 1024 + n_stamp] = (t_ << 8) | (unsigned)(id); ++n_stamp; } } while (0)
#else
#define RR_STAMP(id) do { } while (0)
#endif
template <int MODE, bool IMG = false>
__global__ void __launch_bounds__(RR_THREADS, 2) gemm_rowreg_kernel(GemmParams p)
{
    static_assert(!IMG || MODE >= 1, "the operand image holds fp16 rows");
    constexpr bool F16X2 = MODE >= 1;     // fp16 row operand (MODE 1 and 2)
    constexpr bool F16X1 = MODE == 2;     // ... and a single fp16 query operand
    // (plain constants, not RrGeom<MODE>::X inside the lambdas below: hipcc 7.2 silently drops the HOST-side instantiation of a
    // kernel template whose always_inline lambda names a dependent type alias of the enclosing function -- the stub stays a
    // declaration and the library fails to link)
    constexpr int SLOTS = RrGeom<MODE>::SLOTS, ROW_F4 = RrGeom<MODE>::ROW_F4, SLOT_F4 = RrGeom<MODE>::SLOT_F4;
    constexpr int QUERY_WORDS = RrGeom<MODE>::QUERY_WORDS, AHEAD = RrGeom<MODE>::AHEAD, GT = RrGeom<MODE>::GT;
    constexpr int WAVES = RR_WAVES;
    constexpr int STAGE_ROWS = QT_ROWS / WAVES;      // rows of a query tile each wave stages: 4
    constexpr int STAGE_EVERY = 2;                   // one DMA every so many K-steps at the start of a product
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4 *s_q = reinterpret_cast<f32x4 *>(smem_raw);                  // [4][32][65] float4: query tiles (split image)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    unsigned char *tbuf = smem_raw + SLOTS * SLOT_F4 * 16 + wave * RR_TBUF;
    unsigned char *s_qconst = smem_raw + SLOTS * SLOT_F4 * 16 + RR_WAVES * RR_TBUF;   // [slots][32] (threshold, 1/|q|)
    const uint32_t qs = blockIdx.x % p.qsplit;
    const uint32_t row_block = blockIdx.x / p.qsplit, row_blocks = gridDim.x / p.qsplit;
    const uint32_t qt_lo = (uint32_t)((uint64_t)qs * p.nqt / p.qsplit);
    const uint32_t qt_hi = (uint32_t)((uint64_t)(qs + 1) * p.nqt / p.qsplit);
    const uint32_t n_qt = qt_hi - qt_lo;
    const bool resident = n_qt <= (uint32_t)SLOTS;

    // row u of this wave's share of query tile qt -> LDS slot (padding rows of the image are zero rows: no branch)
    auto stage_row = [&](uint32_t qt, int slot, int u) __attribute__((always_inline)) {
        const int r = wave * STAGE_ROWS + u;  // wave-uniform
        const uint32_t q = qt * QT_ROWS + r;
        f32x4 *dst = s_q + slot * SLOT_F4 + r * ROW_F4;
        if constexpr (F16X1) {   // 512 B per query: the lower half of the wave carries it (every wave still issues ONE instruction)
            if (lane < 32)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(p.queries_split) + (size_t)q * QUERY_WORDS + lane * 4,
                                                 (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(p.queries_split) + (size_t)q * QUERY_WORDS + lane * 4,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    // the tile's 32 (threshold, 1/|q|) pairs: 256 B, one 4-byte DMA.  EVERY wave issues it (same bytes to the same
    // place) so that all waves count the same number of DMA instructions per tile -- the vmcnt arithmetic below
    auto stage_consts = [&](uint32_t qt, int slot) __attribute__((always_inline)) {
        __builtin_amdgcn_global_load_lds(p.qconst + (size_t)qt * QT_ROWS * 2 + lane,
                                         (__attribute__((address_space(3))) void *)(s_qconst + slot * RR_QCONST), 4, 0, 0);
    };
    auto stage_tile = [&](uint32_t qt, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < STAGE_ROWS; ++u) stage_row(qt, slot, u);
        stage_consts(qt, slot);
    };
    auto stage_wait = [&]() { __builtin_amdgcn_s_waitcnt(0x0F70); };  // vmcnt(0)
    {
        const uint32_t first = resident ? n_qt : (uint32_t)((SMT_RR_EXP & 2) ? SLOTS : AHEAD);   // streaming: ring positions 0 .. AHEAD - 1
        for (uint32_t t = 0; t < first; ++t) stage_tile(qt_lo + t, (int)t);
        stage_wait();
    }
    __syncthreads();

    const uint64_t W = (uint64_t)row_blocks * WAVES;
    const uint64_t steps = (p.level_tiles - p.tile_begin + W - 1) / W;  // block-uniform trip count
    uint64_t it = p.tile_begin + (uint64_t)row_block * WAVES + wave;
    uint32_t pos = 0;            // streaming: running ring position (block-uniform); slot = pos & (SLOTS - 1)
    uint32_t tq = 0, tq_ahead = (uint32_t)AHEAD % n_qt;   // tile at position pos / pos + AHEAD (the tile sequence is cyclic over the sweeps)
    // transpose geometry: this lane WRITES row (8u + lane/8), bytes 8 * (lane%8) of a slice; it READS row j, quads 2mm + h
    const uint32_t t_wr = (uint32_t)((lane >> 3) * RR_TROW + (lane & 7) * 8);
    const uint32_t t_rd = (uint32_t)(j * RR_TROW + h * 16);

#if (SMT_RR_EXP & 8)
    u32x4 Ah[16], Al[F16X2 ? 1 : 16];
    unsigned zero16 = 0, valid16 = 0;
#endif
#if (SMT_RR_EXP & 256)
    int n_stamp = 0;
#endif
    uint32_t n_buf = 0;          // nominations waiting in this wave's LDS buffer (wave-uniform)
    for (uint64_t step = 0; step < steps; ++step, it += W) {
        const bool has = it < p.level_tiles;  // wave-uniform
        const uint64_t row0 = (has ? level_tile(it, p.stride, p.skip16) : 0) * 32;
#if (SMT_RR_EXP & 256)
        const bool tracing = p.stamps != nullptr && blockIdx.x == 40 && step == 1 && p.level_tiles > 100000;
#endif
        RR_STAMP(1);   // step start
#if (SMT_RR_EXP & 256)
        // the shader clock under this load: s_memtime against the 100 MHz s_memrealtime, one step apart
        if (p.stamps != nullptr && blockIdx.x == 40 && wave == 0 && lane == 0 && (step == 1 || step == 9) && p.level_tiles > 100000) {
            p.stamps[8 * 1024 + (step == 1 ? 0 : 2)] = __builtin_readcyclecounter();
            p.stamps[8 * 1024 + (step == 1 ? 1 : 3)] = wall_clock64();
        }
#endif

#if !(SMT_RR_EXP & 8)
        u32x4 Ah[16], Al[F16X2 ? 1 : 16];
        unsigned zero16 = 0, valid16 = 0;
#endif
        if constexpr (IMG) {
            // ---- the operands are READY in the corpus' fp16 image (pack_image_kernel wrote them with the arithmetic of the branch
            // below): 16 loads of 1 KiB, quad m of lane l at 16 (64 m + l) in the tile's 16 KiB -- half the bytes of the f32 rows,
            // no norms, no conversion, no transpose.  (Requesting them one step ahead into a second register set was measured:
            // 8-32 queries 0.95 -> 0.94 ms, one query 0.92 -> 0.89, 256-512 queries 3-5 % slower with the spills it brings: not kept.)
            if (has) {
                const uint64_t tile = row0 >> 5;
                const u32x4 *img = reinterpret_cast<const u32x4 *>(p.image) + tile * 1024 + lane;
#pragma unroll
                for (int m = 0; m < 16; ++m) Ah[m] = __builtin_nontemporal_load(img + m * 64);
                RR_STAMP(2);
                if (n_buf) flush_candidates_lds(lane, tbuf, n_buf, p.cand, p.counts);
                const uint32_t zm = p.image_zero[tile] >> (4 * h);   // bit 8u + c: tile row 8u + c + 4h
                zero16 = 0;
                valid16 = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        zero16 |= ((zm >> (8 * u + c)) & 1u) << (4 * u + c);
                        if (row0 + 8 * u + c + 4 * h < p.n_rows) valid16 |= 1u << (4 * u + c);
                    }
            }
        } else
        if (has && (!(SMT_RR_EXP & 8) || step == 0)) {
            // ---- 32 coalesced loads: instruction i = 4s + u covers rows 8u .. 8u+7, dims 32s .. 32s+31 (128 B per row)
            f32x4 R[32];
            {
                // (rebuilt per tile from the SGPR base: as a loop invariant it was the one value hipcc spilled across the sweep)
                uint32_t l7 = (uint32_t)lane & 7u;
                asm volatile("" : "+v"(l7));
                const f32x4 *base = reinterpret_cast<const f32x4 *>(p.corpus) + l7;
                uint64_t rowv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint64_t r = row0 + 8 * u + (lane >> 3);
                    rowv[u] = r < p.n_rows ? r : 0;   // rows past the end: any valid address, masked by valid16
                }
#pragma unroll
                for (int i = 0; i < 32; ++i) R[i] = __builtin_nontemporal_load(base + rowv[i & 3] * 64 + 8 * (i >> 2));
            }
            RR_STAMP(2);   // row loads issued
            // the previous sweep's nominations leave the transpose buffer now: the atomics' round trip hides behind the row loads
            if (n_buf) flush_candidates_lds(lane, tbuf, n_buf, p.cand, p.counts);
            // ---- 1/|row| for the four rows this lane holds pieces of (8 lanes per row)
            float rb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float part = 0.0f;
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {
                    const f32x4 v = R[4 * sl + u];
                    part += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                }
                part += __shfl_xor(part, 1);
                part += __shfl_xor(part, 2);
                part += __shfl_xor(part, 4);
                rb[u] = part == 0.0f ? 0.0f : __frsqrt_rn(part);
            }
            RR_STAMP(3);   // rows arrived, norms done
            // ---- per slice: scale, split, transpose hi then lo through the wave's buffer
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                if constexpr (F16X2) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const f32x4 v = R[4 * sl + u] * (rb[u] * F16X2_ROW_SCALE);
                        *reinterpret_cast<u32x2 *>(tbuf + t_wr + u * 8 * RR_TROW) = (u32x2){f16_pack2(v.x, v.y), f16_pack2(v.z, v.w)};
                    }
                    Ah[2 * sl] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd);
                    Ah[2 * sl + 1] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd + 32);
                } else {
                    uint32_t hi[4][2], lo[4][2];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const f32x4 v = R[4 * sl + u] * rb[u];
                        bf16_split2(v.x, v.y, hi[u][0], lo[u][0]);
                        bf16_split2(v.z, v.w, hi[u][1], lo[u][1]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        *reinterpret_cast<u32x2 *>(tbuf + t_wr + u * 8 * RR_TROW) = (u32x2){hi[u][0], hi[u][1]};
                    Ah[2 * sl] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd);
                    Ah[2 * sl + 1] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd + 32);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        *reinterpret_cast<u32x2 *>(tbuf + t_wr + u * 8 * RR_TROW) = (u32x2){lo[u][0], lo[u][1]};
                    Al[2 * sl] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd);
                    Al[2 * sl + 1] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd + 32);
                }
            }
            // accumulator reg r = 4u + c <-> tile row 8u + i, i = c + 4h; row 8u + i's scale sits in lanes 8i .. 8i + 7 of rb[u]:
            // one ballot per u instead of sixteen LDS permutes (whose hoisted lane indices were what spilled at 256 VGPRs)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned long long zm = __ballot(rb[u] == 0.0f);
                const uint32_t w = h ? (uint32_t)(zm >> 32) : (uint32_t)zm;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    zero16 |= ((w >> (8 * c)) & 1u) << (4 * u + c);
                    if (row0 + 8 * u + c + 4 * h < p.n_rows) valid16 |= 1u << (4 * u + c);
                }
            }
        }

        RR_STAMP(4);   // operands converted
        // K-steps M0 .. M1 - 1 of one (row tile x query tile) product.  While it runs the wave issues its share of the DMA
        // that brings query tile stage_qt into stage_slot (stage: wave-uniform): one row every second K-step, then the constants.
        auto product_part = [&](f32x16 &acc, int slot, const int M0, const int M1, bool stage, uint32_t stage_qt, int stage_slot) __attribute__((always_inline)) {
            // quad of (K-step m, half h): [hi, lo] pairs at 4 m + 2 h (+ 1) in the 1 KiB image; hi only at 2 m + h in the compact one
            constexpr int QS = F16X1 ? 2 : 4;   // quads per K-step in a query row
            const u32x4 *bq = reinterpret_cast<const u32x4 *>(s_q + slot * SLOT_F4 + j * ROW_F4) + (F16X1 ? h : 2 * h);
            const int M_CONSTS = M0 + (M1 - M0 > STAGE_EVERY * STAGE_ROWS ? STAGE_EVERY * STAGE_ROWS : M1 - M0 - 1);
            if constexpr (F16X1) {
                // B quads arrive in GROUPS of four K-steps, double-buffered: wait for group g (an explicit lgkmcnt(0)), THEN
                // issue the four reads of group g + 1, THEN run the four MFMAs of group g -- the reads fly under 128 cycles of
                // this wave's MFMAs and no MFMA waits for a read issued an instruction earlier (hipcc guards a read issued
                // one or two K-steps ahead, as below for the other modes, with lgkmcnt(0) at every second MFMA).
                constexpr int BG = 4;   // (8, and 16 = no overlap inside a product at all, measure the same: 5.84-5.88 ms at 1000 x 10 M)
                u32x4 B[2][BG];
#pragma unroll
                for (int d = 0; d < BG; ++d) B[0][d] = bq[QS * (M0 + d)];
#pragma unroll
                for (int g = 0; g < (M1 - M0) / BG; ++g) {
                    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0), vmcnt / expcnt unconstrained: group g is in registers
                    if (g + 1 < (M1 - M0) / BG) {
#pragma unroll
                        for (int d = 0; d < BG; ++d) B[(g + 1) & 1][d] = bq[QS * (M0 + BG * (g + 1) + d)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int d = 0; d < BG; ++d) {
                        const int m = M0 + BG * g + d;
                        if ((m - M0) % STAGE_EVERY == 0 && (m - M0) / STAGE_EVERY < STAGE_ROWS && stage) stage_row(stage_qt, stage_slot, (m - M0) / STAGE_EVERY);
                        if (m == M_CONSTS && stage) stage_consts(stage_qt, stage_slot);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Ah[m]), __builtin_bit_cast(f16x8, B[g & 1][d]), acc, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // B quads run RR_BDIST K-steps ahead of their MFMAs (sched_barrier: hipcc otherwise sinks every read to its
                // use and each K-step then starts with a full LDS round trip in front of 96 cycles of MFMA)
                // (bf16 x 3 holds 128 operand VGPRs: one K-step of distance keeps it at 256 registers WITHOUT spilling -- with two
                // it spilled three into scratch inside this loop; measured equal otherwise)
                constexpr int BD = MODE == 0 ? 1 : RR_BDIST;
                constexpr int NB = BD + 1;
                u32x4 bh[NB], bl[NB];
#pragma unroll
                for (int d = 0; d < BD; ++d) { bh[d] = bq[QS * (M0 + d)]; bl[d] = bq[QS * (M0 + d) + 1]; }
#pragma unroll
                for (int m = M0; m < M1; ++m) {
                    if (m + BD < M1) {
                        bh[(m - M0 + BD) % NB] = bq[QS * (m + BD)];
                        bl[(m - M0 + BD) % NB] = bq[QS * (m + BD) + 1];
                    }
                    if ((m - M0) % STAGE_EVERY == 0 && (m - M0) / STAGE_EVERY < STAGE_ROWS && stage) stage_row(stage_qt, stage_slot, (m - M0) / STAGE_EVERY);
                    if (m == M_CONSTS && stage) stage_consts(stage_qt, stage_slot);
                    if constexpr (F16X2) acc = mfma_f16x2(Ah[m], bh[(m - M0) % NB], bl[(m - M0) % NB], acc);
                    else acc = mfma_bf16x3(Ah[m], Al[m], bh[(m - M0) % NB], bl[(m - M0) % NB], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        auto product_epilogue = [&](const f32x16 &acc, uint32_t qt, int slot) __attribute__((always_inline)) {
            // (thresholds from LDS, staged with the tile: a global load here would sit in the same in-order vmcnt queue as the
            // tile DMAs, and waiting for it would wait for them)
            const f32x2 qc = *reinterpret_cast<const f32x2 *>(s_qconst + slot * RR_QCONST + j * 8);
#if (SMT_RR_EXP & 1)
            if (acc[0] + acc[5] + acc[10] + acc[15] == 12345.678f)
#endif
            {
#if (SMT_RR_EXP & 256)
                if (p.buffered) append_candidates_lds(acc, zero16, valid16, qt * QT_ROWS + j, qc.x, qc.y, row0, h, lane, tbuf, n_buf, p.cand, p.counts,
                                                      tracing && n_stamp < 1000 ? p.stamps + wave * 1024 : nullptr, n_stamp);
#else
                if (p.buffered) append_candidates_lds(acc, zero16, valid16, qt * QT_ROWS + j, qc.x, qc.y, row0, h, lane, tbuf, n_buf, p.cand, p.counts);
#endif
                else append_candidates(acc, zero16, valid16, qt * QT_ROWS + j, qc.x, qc.y, row0, h, p.cand, p.counts);
            }
        };
        auto tile_product = [&](uint32_t qt, int slot, bool stage, uint32_t stage_qt, int stage_slot) __attribute__((always_inline)) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            RR_STAMP(5);   // product start
            product_part(acc, slot, 0, 16, stage, stage_qt, stage_slot);
            RR_STAMP(6);   // MFMAs issued
            product_epilogue(acc, qt, slot);
            RR_STAMP(7);   // epilogue done
        };

        if (resident) {
            if (has)
                for (uint32_t t = 0; t < n_qt; ++t) tile_product(qt_lo + t, (int)t, false, 0, 0);
        } else {
            for (uint32_t t = 0; t < n_qt; ++t) {
                const int slot = (int)(pos & (SLOTS - 1)), slot_ahead = (int)((pos + AHEAD) & (SLOTS - 1));   // slot_ahead was read during step pos - 1
#if (SMT_RR_EXP & 2)
                if (has) tile_product(qt_lo + tq, slot, false, qt_lo + tq_ahead, slot_ahead);
#else
                if (has) tile_product(qt_lo + tq, slot, true, qt_lo + tq_ahead, slot_ahead);
                else stage_tile(qt_lo + tq_ahead, slot_ahead);
#endif
                // this wave's share of the next group's tiles has landed: only the younger DMAs may still fly
                // (a raw s_barrier: __syncthreads() carries a fence that hipcc lowers to vmcnt(0), i.e. it would wait
                // for the tiles that are meant to stay in flight.  The LDS reads of this step were consumed by MFMAs.)
                // At the border in front of position pos + 1 the tiles pos + 1 .. pos + GT must have landed; the younger ones,
                // pos + GT + 1 .. pos + AHEAD, may still fly.
                constexpr int FLY = (AHEAD - GT) * (STAGE_ROWS + 1);               // 10 (four slots, GT 1) / 20 (eight, GT 2)
                static_assert(FLY >= 0 && FLY < 64, "vmcnt is a 6-bit counter");
                if ((pos + 1) % GT == 0 && !(SMT_RR_EXP & 4)) {   // block-uniform
                    __builtin_amdgcn_s_waitcnt(0x0F70 | (FLY & 15) | ((FLY >> 4) << 14));  // vmcnt(FLY), expcnt / lgkmcnt unconstrained
                    RR_STAMP(8);   // own DMAs landed
                    __builtin_amdgcn_s_barrier();
                    RR_STAMP(9);   // barrier passed
                }
                asm volatile("" ::: "memory");
                ++pos;
                tq = tq + 1 == n_qt ? 0 : tq + 1;
                tq_ahead = tq_ahead + 1 == n_qt ? 0 : tq_ahead + 1;
            }
        }
    }
    if (n_buf) flush_candidates_lds(lane, tbuf, n_buf, p.cand, p.counts);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): no LDS-DMA may outlive the block's LDS allocation
}

// ---- small / medium batches on f32 MFMAs and every range-filtered batch (nq <= 64 per pass): ROW TILES LAND IN LDS
// BY LDS-DMA.  (The first-generation kernel of this regime, removed since, pulled its 32-row tile straight into
// registers with fragment-shaped loads: one wave instruction touches 32 B of 32 different rows -- 32 queries x 10 M
// rows took 3.0 ms against an HBM bound of 1.28 ms.  gemm_rowreg_kernel above is the third answer to the same
// problem and the default for unfiltered batches.)  Here
//   * the corpus streams through a per-wave LDS ring of 8 KiB K-SLICES (32 rows x 64 dims) filled by
//     global_load_lds_dwordx4: one instruction moves 4 rows x 256 contiguous bytes, no staging registers, no ds_write;
//   * A fragments are read from the ring with ds_read_b128 (one read feeds 4 x NQT MFMAs); a slot is refilled the
//     moment its fragments sit in registers, so the ring only ever holds bytes in flight;
//   * NQT = 1 (<= 32 queries, HBM-bound): the query tile is the register-resident MFMA operand (128 VGPRs), the
//     ring has two slots per wave (16 KiB/wave, 128 KiB/CU in flight), counted s_waitcnt vmcnt(8);
//     NQT = 2 (<= 64 queries, MFMA-bound): both operands come from LDS (64 KiB of swizzled queries + one slot per
//     wave), the 64 MFMAs of a slice cover the refill of its slot;
//   * EIGHT waves per CU, two per SIMD (<= 256 VGPRs): the first version ran one wave per SIMD and measured
//     MFMA time + everything else, back to back (ablations on MI355X, 32 queries x 10 M rows: MFMAs alone 1.04 ms,
//     LDS reads + address work alone 0.63 ms, together 1.68 ms; epilogue +0.25 ms; DMA waits +0.25 ms) -- a lone
//     wave issues in order, so its own LDS waits, norm FMAs and epilogue stall its MFMA stream; the second wave of
//     the SIMD fills those holes (gemm_level_kernel already worked that way);
//   * no barrier anywhere: every wave runs its own pipeline (s_waitcnt vmcnt(N) covers the issuing wave's LDS-DMA).
// LDS image: rows are 256 B apart inside a slice (no padding: LDS-DMA writes lane-linear), so the 16-B chunk c of
// row i is stored at position c ^ (i & 7): the swizzle is applied to the SOURCE address of the DMA and to the
// fragment read (same involution on both sides), which spreads the 8 lanes of a read phase over all 32 banks.
// Row norms come from the fragments (each lane squares the half row it reads anyway); the scale 1/|row| moves to
// the epilogue (acc * rb >= threshold) and travels through a 128-B LDS scratch (one write, four b128 reads per lane
// instead of 16 ds_bpermute round trips).
// FILTERED: tile t = chunks 8t .. 8t+7 of the chunk table (4 rows each, the last of a range short): DMA instruction
// u of a slice covers exactly chunk u, whose descriptor is a wave-uniform scalar load.
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

// LDS-DMA of one K-slice (dims 64*S ..) of a tile into `slot`: 8 instructions, each moves 4 rows x 256 B (1 KiB,
// lane-linear in LDS).  rows[u] = this lane's row for instruction u; swz = its swizzled chunk offset in floats for
// even / odd u.  (A free function template, not a generic lambda inside the kernel: with
// __builtin_amdgcn_global_load_lds inside a generic lambda hipcc 7.2 silently drops the kernel's HOST stub.)
template <int S, int AUX>
__device__ __forceinline__ void lr_fill_slice(const float *corpus, const uint32_t (&rows)[8], uint32_t swz_even, uint32_t swz_odd,
                                              unsigned char *slot)
{
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const float *src = corpus + (uint64_t)rows[u] * 256 + ((u & 1) ? swz_odd : swz_even) + 64 * S;
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(slot + u * 1024), 16, 0, AUX);
    }
}

// AUX: cache-policy bits of the row DMA (0 = default, 2 = nt: the corpus is streamed once)
// BF16: bf16 x 3 split products (mfma_tile.h).  The rows still arrive as f32 (LDS-DMA moves bytes); a wave splits the
// fragments it reads (2.5 VALU instructions per element, next to 3 MFMAs of 32 cycles per 8 elements instead of 8 MFMAs
// of 64) -- the MFMA pipe drops from 55 % busy to 10 % at 32 queries and the kernel is purely a question of row arrival.
template <int NQT, bool FILTERED, int AUX, bool BF16>
__global__ void __launch_bounds__(LR_THREADS) gemm_ldsrow_kernel(GemmParams p)
{
    using G = LrGeom<NQT>;
    constexpr bool B_REGS = NQT == 1;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    unsigned char *ring = smem_raw + G::Q_BYTES + wave * G::RING_BYTES;
    float *scale = reinterpret_cast<float *>(smem_raw + G::SCRATCH_OFF + wave * 128);

    // fragment geometry: lane (j, h) reads chunk 2m'+h of row j, stored at position chunk ^ (j & 7)
    // (BF16: K-step m'' of the slice takes chunks 4m'' + 2h and + 1: fragments 2m'', 2m'' + 1)
    uint32_t foff[8];  // byte offset of fragment m' inside a slice
#pragma unroll
    for (int mp = 0; mp < 8; ++mp) {
        const int chunk = BF16 ? 4 * (mp >> 1) + 2 * h + (mp & 1) : 2 * mp + h;
        foff[mp] = (uint32_t)(j * 256 + ((chunk ^ (j & 7)) << 4));
    }

    // ---- B operand: this lane's query of every tile, K-permuted like the A fragments (dims 8m+4h .. +3 in group m)
    f32x4 Bq[B_REGS && !BF16 ? 32 : 1];
    u32x4 Bh[B_REGS && BF16 ? 16 : 1], Bl[B_REGS && BF16 ? 16 : 1];
    float thr[NQT], rq[NQT];
    if constexpr (!B_REGS) {
        // queries -> LDS by LDS-DMA, one 1 KiB row per instruction, chunk c of row r at position c ^ (r & 7)
        // (BF16: the rows of the split image, see split_queries_kernel)
        const float *qsrc = BF16 ? reinterpret_cast<const float *>(p.queries_split) : p.queries;
        for (int r = wave; r < NQT * QT_ROWS; r += LR_WAVES) {  // wave-uniform
            unsigned char *dst = smem_raw + r * 1024;
            if ((uint32_t)r < p.nq)
                __builtin_amdgcn_global_load_lds(qsrc + (size_t)r * 256 + ((lane ^ (r & 7)) << 2),
                                                 (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            else
                reinterpret_cast<f32x4 *>(dst)[lane] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __syncthreads();                     // the only barrier: the query image is shared by the block's waves
    }
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        const uint32_t q = t * QT_ROWS + j;
        const bool ok = q < p.nq;
        float part = 0.0f;
        if constexpr (BF16) {
            // the norm always comes from the f32 query; with B in registers the lane also splits its operand quads
            const f32x4 *src = reinterpret_cast<const f32x4 *>(p.queries + (size_t)(ok ? q : 0) * 256) + 2 * h;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                f32x4 v0 = src[4 * m], v1 = src[4 * m + 1];
                if (!ok) v0 = v1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                part += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
                part += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
                if constexpr (B_REGS) bf16_split8(v0, v1, Bh[m], Bl[m]);
            }
        } else if constexpr (B_REGS) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(p.queries + (size_t)(ok ? q : 0) * 256) + h;
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                f32x4 v = src[2 * m];
                if (!ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                Bq[m] = v;
                part += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        } else {
            const unsigned char *qrow = smem_raw + t * LR_QTILE_BYTES + j * 1024;
#pragma unroll 8
            for (int m = 0; m < 32; ++m) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(qrow + (((2 * m + h) ^ (j & 7)) << 4));
                part += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        }
        const float a2 = part + __shfl_xor(part, 32);
        rq[t] = a2 == 0.0f ? 0.0f : __frsqrt_rn(a2);
        thr[t] = score_threshold(ok ? p.tau[q] : -1.0f, rq[t]);  // padding: zero query, tau < 0 -> never passes
    }
    // (every ordinary global load above has been consumed: none is pending when the first row DMA is issued)

    const uint64_t W = (uint64_t)gridDim.x * LR_WAVES;
    uint64_t it = (uint64_t)blockIdx.x * LR_WAVES + wave;
    if (it >= p.level_tiles) return;
    const const_u64_ptr table = (const_u64_ptr)(uintptr_t)p.chunk_table;

    // lane geometry of one DMA instruction: rows 4u .. 4u+3 of the tile, 256 B of each
    const int rl = lane >> 4, pos = lane & 15;
    const uint32_t swz_even = (uint32_t)((pos ^ rl) << 2);        // tile row i = 4u + rl: i & 7 = rl (u even)
    const uint32_t swz_odd = (uint32_t)((pos ^ (4 + rl)) << 2);   //                             4 + rl (u odd)

    struct TileSrc {
        uint32_t rows[8];      // per DMA instruction: this lane's corpus row
        uint32_t row0[8];      // FILTERED: first row of chunk u (wave-uniform)
        uint32_t valid32;      // bit i: tile row i exists (wave-uniform)
        uint32_t first_row;    // unfiltered: row of tile row 0
    };
    auto describe = [&](uint64_t tile, TileSrc &d) {
        d.valid32 = 0;
        d.first_row = (uint32_t)(tile * 32);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (FILTERED) {
                const uint64_t c = tile * 8 + u;
                const uint64_t desc = c < p.n_chunks ? table[c] : 0ull;  // wave-uniform: scalar load
                const uint32_t r0 = (uint32_t)desc, cnt = (uint32_t)(desc >> 32);
                d.row0[u] = r0;
                d.valid32 |= ((1u << cnt) - 1u) << (4 * u);
                d.rows[u] = r0 + ((uint32_t)rl < cnt ? (uint32_t)rl : (cnt ? cnt - 1 : 0u));
            } else {
                d.row0[u] = 0;
                const uint64_t row = tile * 32 + 4 * u + rl;
                d.rows[u] = (uint32_t)(row < p.n_rows ? row : p.n_rows - 1);  // clamp: fetched, never used (valid32)
            }
        }
        if constexpr (!FILTERED) {
            const uint64_t left = p.n_rows > tile * 32 ? p.n_rows - tile * 32 : 0;
            d.valid32 = left >= 32 ? 0xFFFFFFFFu : ((1u << (uint32_t)left) - 1u);
        }
    };
    auto fill = [&](const TileSrc &d, auto S, int slot) {
        lr_fill_slice<decltype(S)::value, AUX>(p.corpus, d.rows, swz_even, swz_odd, ring + slot * LR_SLICE_BYTES);
    };
    auto read_frags = [&](int slot, f32x4 (&f)[8]) {
#pragma unroll
        for (int mp = 0; mp < 8; ++mp) f[mp] = *reinterpret_cast<const f32x4 *>(ring + slot * LR_SLICE_BYTES + foff[mp]);
    };
    // counted waits (imm: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14)
    auto wait_landed = [&]() {   // the NEXT slice has landed; with two slots 8 younger instructions may still fly
        if constexpr (G::SLOTS == 2) __builtin_amdgcn_s_waitcnt(0x0F78);  // vmcnt(8)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0)
        asm volatile("" ::: "memory");
    };
    auto wait_lgkm0 = [&]() { __builtin_amdgcn_s_waitcnt(0xC07F); asm volatile("" ::: "memory"); };  // lgkmcnt(0), vmcnt untouched

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    TileSrc cur_t, nxt_t;
    describe(level_tile(it, p.stride, p.skip16), cur_t);
    fill(cur_t, S0{}, 0);
    if constexpr (G::SLOTS == 2) fill(cur_t, S1{}, 1);
    wait_landed();
    f32x4 fr[8];
    read_frags(0, fr);

    for (; it < p.level_tiles; it += W) {
        const bool more = it + W < p.level_tiles;  // wave-uniform
        if (more) describe(level_tile(it + W, p.stride, p.skip16), nxt_t);
        else nxt_t = cur_t;                        // dummy refills keep the vmcnt arithmetic uniform (valid addresses)

        f32x16 acc[NQT];
#pragma unroll
        for (int t = 0; t < NQT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        float part = 0.0f;

        // One K-slice.  `fr` holds slice S's fragments (its slot is free): refill that slot with the slice SLOTS
        // ahead, multiply, then pull the next slice's fragments into `fr`.
        auto step = [&](auto S, auto SAHEAD, const TileSrc &ahead_tile) {
            constexpr int s = decltype(S)::value;
            constexpr int slot = G::SLOTS == 2 ? (s & 1) : 0;
            constexpr int next_slot = G::SLOTS == 2 ? ((s + 1) & 1) : 0;
            wait_lgkm0();                              // the ds_reads of `fr` have returned: its slot is free
            fill(ahead_tile, SAHEAD, slot);
            if constexpr (BF16) {
#pragma unroll
                for (int mq = 0; mq < 4; ++mq) {       // K-step 4s + mq
                    const f32x4 a0 = fr[2 * mq], a1 = fr[2 * mq + 1];
                    part += a0.x * a0.x + a0.y * a0.y + a0.z * a0.z + a0.w * a0.w;
                    part += a1.x * a1.x + a1.y * a1.y + a1.z * a1.z + a1.w * a1.w;
                    u32x4 ah, al;
                    bf16_split8(a0, a1, ah, al);
#pragma unroll
                    for (int t = 0; t < NQT; ++t) {
                        u32x4 bh, bl;
                        if constexpr (B_REGS) {
                            bh = Bh[4 * s + mq];
                            bl = Bl[4 * s + mq];
                        } else {
                            const unsigned char *qrow = smem_raw + t * LR_QTILE_BYTES + j * 1024;
                            const int c = 2 * (2 * (4 * s + mq) + h);  // chunk of the hi quad; lo is the next one
                            bh = *reinterpret_cast<const u32x4 *>(qrow + ((c ^ (j & 7)) << 4));
                            bl = *reinterpret_cast<const u32x4 *>(qrow + (((c + 1) ^ (j & 7)) << 4));
                        }
                        acc[t] = mfma_bf16x3(ah, al, bh, bl, acc[t]);
                    }
                }
            } else
#pragma unroll
            for (int mp = 0; mp < 8; ++mp) {
                const f32x4 a = fr[mp];
                part += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
#pragma unroll
                for (int t = 0; t < NQT; ++t) {
                    f32x4 b;
                    if constexpr (B_REGS) b = Bq[8 * s + mp];
                    else b = *reinterpret_cast<const f32x4 *>(smem_raw + t * LR_QTILE_BYTES + j * 1024 +
                                                              (((2 * (8 * s + mp) + h) ^ (j & 7)) << 4));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);         // the MFMAs stay above: `fr` is single-buffered (register budget)
            wait_landed();                             // the next slice is in LDS
            read_frags(next_slot, fr);
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (G::SLOTS == 2) {
            step(S0{}, S2{}, cur_t);   // slot 0 <- (this tile, slice 2)
            step(S1{}, S3{}, cur_t);   // slot 1 <- (this tile, slice 3)
            step(S2{}, S0{}, nxt_t);   // slot 0 <- (next tile, slice 0)
            step(S3{}, S1{}, nxt_t);   // slot 1 <- (next tile, slice 1); leaves the next tile's slice-0 fragments in fr
        } else {
            step(S0{}, S1{}, cur_t);
            step(S1{}, S2{}, cur_t);
            step(S2{}, S3{}, cur_t);
            step(S3{}, S0{}, nxt_t);
        }

        // ---- epilogue: lane (j, h) owns query j of every tile and the 16 rows acc_row(r, h)
        const float b2 = part + __shfl_xor(part, 32);   // row j's norm^2 (both halves hold it)
        const float rb = b2 == 0.0f ? 0.0f : __frsqrt_rn(b2);
        if (h == 0) scale[j] = rb;                      // wave-private scratch: no barrier, an lgkmcnt wait orders it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float rbv[16];
        unsigned zero16 = 0;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + 8 * gq + 4 * h);  // rows acc_row(4gq .. 4gq+3, h)
            rbv[4 * gq + 0] = sc.x; rbv[4 * gq + 1] = sc.y; rbv[4 * gq + 2] = sc.z; rbv[4 * gq + 3] = sc.w;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (rbv[r] == 0.0f) zero16 |= 1u << r;
        const uint32_t v = cur_t.valid32 >> (4 * h);
        const unsigned valid16 = (v & 0xFu) | (((v >> 8) & 0xFu) << 4) | (((v >> 16) & 0xFu) << 8) | (((v >> 24) & 0xFu) << 12);
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            const uint32_t q = t * QT_ROWS + j;
            unsigned pass = 0;
            auto dist_of = [&](int r) {
                if (rq[t] == 0.0f) return (zero16 >> r) & 1u ? 0.0f : 1.0f;   // zero query (simsimd rules)
                return fmaxf(1.0f - acc[t][r] * rbv[r] * rq[t], 0.0f);         // a zero row has rb == 0 -> 1
            };
            if (rq[t] != 0.0f) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (acc[t][r] * rbv[r] >= thr[t]) pass |= 1u << r;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (dist_of(r) <= thr[t]) pass |= 1u << r;
            }
            pass &= valid16;
            if (__builtin_amdgcn_ballot_w64(pass != 0)) {
                if (pass) {
                    const unsigned base = atomicAdd(&p.counts[q], (unsigned)__popc(pass));
                    key_t64 *dst = p.cand + (size_t)q * CAND_CAP;
                    unsigned slot = base;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (pass & (1u << r)) {
                            const int i = acc_row(r, h);
                            // (constant indices only: a lane-dependent index would send row0[] to scratch)
                            const uint32_t r0 = h ? cur_t.row0[2 * (r >> 2) + 1] : cur_t.row0[2 * (r >> 2)];
                            const uint32_t row = FILTERED ? r0 + (uint32_t)(i & 3) : cur_t.first_row + (uint32_t)i;
                            if (slot < CAND_CAP) dst[slot] = make_key(dist_of(r), row);
                            ++slot;
                        }
                    }
                }
            }
        }
        cur_t = nxt_t;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): no LDS-DMA may outlive the wave's LDS allocation
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

__global__ void __launch_bounds__(1024) level_select_kernel(LevelSelectParams p)
{
    __shared__ key_t64 s_keys[CAND_CAP];
    __shared__ key_t64 s_best[64];
    const uint32_t q = blockIdx.x;
    key_t64 *buf = p.cand + (size_t)q * CAND_CAP;
    unsigned n = p.counts[q];
    if (n > CAND_CAP) {
        if (threadIdx.x == 0) p.overflow[q] = 1;
        n = CAND_CAP;
    }
    for (unsigned e = threadIdx.x; e < n; e += blockDim.x) s_keys[e] = buf[e];
    if (threadIdx.x < 64) s_best[threadIdx.x] = KEY_PAD;
    __syncthreads();
    for (unsigned e = threadIdx.x; e < n; e += blockDim.x) {
        const key_t64 key = s_keys[e];
        unsigned rank = 0;
        for (unsigned i = 0; i < n; ++i) rank += (s_keys[i] < key) ? 1u : 0u;
        if (rank < p.kp) s_best[rank] = key;
    }
    __syncthreads();
    if (threadIdx.x < p.kp) buf[threadIdx.x] = s_best[threadIdx.x];
    if (threadIdx.x == 0) {
        p.counts[q] = n < p.kp ? n : p.kp;
        const float tau = n >= p.kp ? __uint_as_float((unsigned)(s_best[p.kp - 1] >> 32)) : __builtin_inff();
        p.tau[q] = tau;
        if (p.qconst) p.qconst[2 * q] = score_threshold(tau, p.qconst[2 * q + 1]);
    }
}

__global__ void fill_f32_kernel(float *p, float v, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- test hook: the NOMINATING distances themselves (never part of an answer).  One wave per 32-row tile against one
// tile of <= 32 queries, the same operand preparation and MFMA sequence as gemm_level_kernel; out[row][32] = the f32
// distance the candidate test sees.  tests/test_gpu_batched.py measures |out - exact| against F32_ERR_MFMA / _BF16X3.
template <int MODE>   // 0 f32 MFMA, 1 bf16 x 3, 2 f16 x 2, 3 f16 x 1
__global__ void __launch_bounds__(64) gemm_debug_scores_kernel(const float *corpus, uint64_t first_row, uint32_t n_rows,
                                                               const float *queries, uint32_t nq, float *out)
{
    const int lane = threadIdx.x, h = lane >> 5, j = lane & 31;
    const uint64_t row = first_row + (uint64_t)blockIdx.x * 32 + j;
    const bool row_ok = (uint64_t)blockIdx.x * 32 + j < n_rows;
    const f32x4 *rsrc = reinterpret_cast<const f32x4 *>(corpus + (row_ok ? row : first_row) * 256);
    const f32x4 *qsrc = reinterpret_cast<const f32x4 *>(queries + (size_t)((uint32_t)j < nq ? j : 0) * 256);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float rpart = 0.0f, qpart = 0.0f;
    f32x4 R[32], Q[32];
    // f32: lane (j, h) feeds dims 8m + 4h .. + 3 of group m; bf16: dims 16m + 8h .. + 7 of K-step m
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        const int idx = MODE ? 4 * (m >> 1) + 2 * h + (m & 1) : 2 * m + h;
        R[m] = row_ok ? rsrc[idx] : (f32x4){0.f, 0.f, 0.f, 0.f};
        Q[m] = (uint32_t)j < nq ? qsrc[idx] : (f32x4){0.f, 0.f, 0.f, 0.f};
        rpart += R[m].x * R[m].x + R[m].y * R[m].y + R[m].z * R[m].z + R[m].w * R[m].w;
        qpart += Q[m].x * Q[m].x + Q[m].y * Q[m].y + Q[m].z * Q[m].z + Q[m].w * Q[m].w;
    }
    const float r2 = rpart + __shfl_xor(rpart, 32), q2 = qpart + __shfl_xor(qpart, 32);
    const float rb = r2 == 0.0f ? 0.0f : __frsqrt_rn(r2);
    float rq = q2 == 0.0f ? 0.0f : __frsqrt_rn(q2);
    if constexpr (MODE >= 2) {
        const float qs = rq * F16X2_QUERY_SCALE;   // unit query x 2^8, unit row x 2^10 (as split_queries_f16_kernel / gemm_rowreg_kernel<true>)
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const f32x4 r0 = R[2 * m] * (rb * F16X2_ROW_SCALE), r1 = R[2 * m + 1] * (rb * F16X2_ROW_SCALE);
            const f32x4 q0 = Q[2 * m] * qs, q1 = Q[2 * m + 1] * qs;
            const u32x4 a = {f16_pack2(r0.x, r0.y), f16_pack2(r0.z, r0.w), f16_pack2(r1.x, r1.y), f16_pack2(r1.z, r1.w)};
            uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
            f16_split2(q0.x, q0.y, h0, l0);
            f16_split2(q0.z, q0.w, h1, l1);
            f16_split2(q1.x, q1.y, h2, l2);
            f16_split2(q1.z, q1.w, h3, l3);
            if constexpr (MODE == 3)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, (u32x4){h0, h1, h2, h3}), acc, 0, 0, 0);
            else acc = mfma_f16x2(a, (u32x4){h0, h1, h2, h3}, (u32x4){l0, l1, l2, l3}, acc);
        }
        if (rq != 0.0f) rq = F16X2_INV_SCALE;
    } else if constexpr (MODE == 1) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            u32x4 ah, al, bh, bl;
            bf16_split8(R[2 * m] * rb, R[2 * m + 1] * rb, ah, al);
            bf16_split8(Q[2 * m], Q[2 * m + 1], bh, bl);
            acc = mfma_bf16x3(ah, al, bh, bl, acc);
        }
    } else {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const f32x4 a = R[m] * rb, b = Q[m];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t i = blockIdx.x * 32 + acc_row(r, h);
        if (i < n_rows) out[(size_t)i * 32 + j] = fmaxf(1.0f - acc[r] * rq, 0.0f);
    }
}

int launch_gemm_debug_scores(smt_ctx *ctx, const float *corpus, uint64_t first_row, uint32_t n_rows, const float *queries,
                             uint32_t nq, float *out)
{
    if (nq < 1 || nq > 32 || n_rows < 1) { set_error("debug scores: 1..32 queries, >= 1 row"); return SMT_E_INVALID; }
    const dim3 grid((n_rows + 31) / 32);
    if (ctx->tune.gemm_bf16x3 && ctx->tune.gemm_nominate == 3) hipLaunchKernelGGL(gemm_debug_scores_kernel<3>, grid, dim3(64), 0, ctx->stream, corpus, first_row, n_rows, queries, nq, out);
    else if (ctx->tune.gemm_bf16x3 && ctx->tune.gemm_nominate == 2) hipLaunchKernelGGL(gemm_debug_scores_kernel<2>, grid, dim3(64), 0, ctx->stream, corpus, first_row, n_rows, queries, nq, out);
    else if (ctx->tune.gemm_bf16x3) hipLaunchKernelGGL(gemm_debug_scores_kernel<1>, grid, dim3(64), 0, ctx->stream, corpus, first_row, n_rows, queries, nq, out);
    else hipLaunchKernelGGL(gemm_debug_scores_kernel<0>, grid, dim3(64), 0, ctx->stream, corpus, first_row, n_rows, queries, nq, out);
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

template <int NQT, bool FILTERED, int AUX, bool BF16>
static hipError_t lr_attr()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_ldsrow_kernel<NQT, FILTERED, AUX, BF16>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <bool BF16>
static hipError_t lr_set_attr()
{
    hipError_t e;
    if ((e = lr_attr<1, false, 0, BF16>()) != hipSuccess) return e;
    if ((e = lr_attr<2, false, 0, BF16>()) != hipSuccess) return e;
    if ((e = lr_attr<1, false, 2, BF16>()) != hipSuccess) return e;
    if ((e = lr_attr<2, false, 2, BF16>()) != hipSuccess) return e;
    if ((e = lr_attr<1, true, 0, BF16>()) != hipSuccess) return e;
    return lr_attr<2, true, 0, BF16>();
}
template <int NQT, bool FILTERED, int AUX, bool BF16>
static void lr_launch(smt_ctx *ctx, int nb, size_t smem, const GemmParams &g)
{
    hipLaunchKernelGGL((gemm_ldsrow_kernel<NQT, FILTERED, AUX, BF16>), dim3(nb), dim3(LR_THREADS), smem, ctx->stream, g);
}
template <bool BF16>
static void lr_dispatch(smt_ctx *ctx, uint32_t nqt, bool filtered, bool nt, int nb, size_t smem, const GemmParams &g)
{
    if (nqt <= 1) {
        if (filtered) lr_launch<1, true, 0, BF16>(ctx, nb, smem, g);
        else if (nt) lr_launch<1, false, 2, BF16>(ctx, nb, smem, g);
        else lr_launch<1, false, 0, BF16>(ctx, nb, smem, g);
    } else {
        if (filtered) lr_launch<2, true, 0, BF16>(ctx, nb, smem, g);
        else if (nt) lr_launch<2, false, 2, BF16>(ctx, nb, smem, g);
        else lr_launch<2, false, 0, BF16>(ctx, nb, smem, g);
    }
}

static size_t gemm_smem_bytes(uint32_t nqt)
{
    return (size_t)4 * QT_F4 * 16 + (size_t)nqt * QT_ROWS * 4 * 2 + 64;
}

static int ensure_gemm_attrs(smt_ctx *ctx)
{
    if (!(ctx->attr_done & ATTR_GEMM)) {  // per context == per device
        SMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_level_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_level_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_rowreg_kernel<0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_rowreg_kernel<1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_rowreg_kernel<2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_rowreg_kernel<1, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SMT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_rowreg_kernel<2, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SMT_HIP_CHECK((lr_set_attr<false>()));
        SMT_HIP_CHECK((lr_set_attr<true>()));
        ctx->attr_done |= ATTR_GEMM;
    }
    return SMT_OK;
}

int launch_gemm_topk(smt_ctx *ctx, const ScanArgs &a)
{
    if (a.k_out + 8 > 64 || a.k_out < 1) { set_error("batched path: top_k must be in [1, 56]"); return SMT_E_UNSUPPORTED; }
    SMT_REQUIRE(a.rows < 0xFFFFFFFFull, "a shard holds fewer than 2^32-1 rows");
    const bool filtered = a.n_ranges != 0;
    const uint32_t nqt = (a.nq + QT_ROWS - 1) / QT_ROWS;
    const uint64_t ostride = a.out_stride ? a.out_stride : a.k_out;
    // Which kernel (measured on MI355X, 10 M rows, ms per batch: LDS-row kernel / gemm_level_kernel):
    //   8..32 queries 1.84 / 3.1;  64 queries 3.05 / 3.24;  96 queries 4.96 (two passes) / 4.32;  128: 6.04 / 5.48.
    // So: up to 64 queries, and every range-filtered batch (in passes of 64), take the LDS-row kernel; larger batches
    // stream the query tiles through LDS (gemm_level_kernel), one sweep over the corpus for up to 3584 queries.
    // bf16 x 3 (the default): gemm_rowreg_kernel takes every unfiltered batch; range-filtered batches keep the LDS-row
    // kernel (its chunk table gathers the rows).  f32 MFMA (gemm_bf16x3 = 0): the round-1/2 routing below.
    const bool rowreg = ctx->tune.gemm_bf16x3 && ctx->tune.gemm_rowreg && !filtered;
    // How gemm_rowreg_kernel nominates (tuning key gemm_nominate: 0 auto, 1 bf16 x 3, 2 f16 x 2, 3 f16 x 1).  The fp16 modes
    // issue 2/3 resp. 1/3 of the MFMAs of bf16 x 3 -- what large batches are bound by -- for a wider certificate band
    // (5.2e-4 / 1.0e-3 against 7e-5): auto takes f16 x 2 from 128 queries and f16 x 1 from 256 queries on shards of at most
    // 32 M rows (the rank spacing of the distances shrinks with the shard; at 10 M random rows the k-th and k+8-th distances
    // are ~8e-3 apart), provided the lists have room for the wider guard band.  Small batches are HBM-bound: bf16 x 3.
    // ... and on larger shards when the operand image is there: at 100 M rows one query takes 7.6 ms from the image against
    // 14.2 ms from the f32 rows, 64 queries 8.9 against 18.7, no query without its certificate (profiles/r03_image_scan_100M.json;
    // a corpus full of near-duplicates pays with exhaustive re-answers instead -- guard_band).
    const bool auto_fp16 = rowreg && ctx->tune.gemm_nominate == 0 &&
                           (a.rows <= (1ull << 25) || (a.image != nullptr && ctx->tune.gemm_image != 0 && a.rows <= (1ull << 28)));
    // With the corpus' fp16 operand image at hand (ScanArgs::image) the fp16 modes read HALF the bytes per row and skip the row
    // phase: f16 x 2 then also takes the batches below 128 queries, which are HBM-bound.
    const bool have_image = a.image != nullptr && rowreg && ctx->tune.gemm_image != 0;
    // (measured with the image, 10 M rows, ms: 128 queries f16 x 1 1.16 / f16 x 2 1.32; 192: 1.23 / 2.09 -- six tiles no longer
    // fit the four slots of the 1 KiB query image; <= 96: equal)
    const bool f16x1 = rowreg && (ctx->tune.gemm_nominate == 3 ||
                                  (auto_fp16 && (nqt >= 8 || (have_image && nqt >= 4)) && a.k_out + 24 <= 64));
    const bool f16x2 = rowreg && !f16x1 && (ctx->tune.gemm_nominate == 2 || (auto_fp16 && (nqt >= 4 || (have_image && a.k_out + 16 <= 64))));
    const bool use_image = have_image && (f16x1 || f16x2);
    // guard band, see candidates_per_list (scan_kernels.hip): the wider the certificate band, the more rows are nominated
    // (the proof needs the k-th exact distance to lie 2 x the band below the worst nominated one): 8 / 16 / 24
    const uint32_t kp = std::min<uint32_t>(64, a.k_out + (uint32_t)std::max(ctx->tune.guard_band, f16x1 ? 24 : f16x2 ? 16 : 8));
    const bool lds_rows = !rowreg && ctx->tune.gemm_ldsrow && (filtered || nqt <= 2);
    if (filtered && !lds_rows) { set_error("range-filtered batches need the LDS-row kernel (tuning key gemm_ldsrow)"); return SMT_E_UNSUPPORTED; }
    const uint32_t pass_nq = lds_rows ? 2 * QT_ROWS : GEMM_MAX_NQ;
    if (a.nq > pass_nq) {
        // (GEMM_MAX_NQ: the per-query thresholds of one gemm_level_kernel launch live in LDS beside the four
        // query-tile slots: larger batches are answered in chunks, each its own sweep over the corpus)
        for (uint32_t q0 = 0; q0 < a.nq; q0 += pass_nq) {
            ScanArgs c = a;
            c.nq = std::min<uint32_t>(pass_nq, a.nq - q0);
            c.queries = a.queries + (size_t)q0 * 256;
            c.out_rows = a.out_rows + (size_t)q0 * ostride;
            c.out_dist = a.out_dist + (size_t)q0 * ostride;
            c.out_counts = a.out_counts ? a.out_counts + q0 : nullptr;
            c.out_uncertain = a.out_uncertain ? a.out_uncertain + q0 : nullptr;
            const int rc_chunk = launch_gemm_topk(ctx, c);
            if (rc_chunk) return rc_chunk;
        }
        return SMT_OK;
    }
    if (gemm_smem_bytes(nqt) > 160 * 1024) { set_error("batch too large for one launch"); return SMT_E_UNSUPPORTED; }

    if (int rc_attr = ensure_gemm_attrs(ctx)) return rc_attr;

    // scratch: cand [nq][CAP] keys | counts [nq] | overflow [nq] | tau [nqt*32] | split queries [nqt*32][1 KiB] | chunk table
    const bool bf16 = ctx->tune.gemm_bf16x3 != 0;
    const size_t b_cand = (size_t)a.nq * CAND_CAP * sizeof(key_t64);
    const size_t b_cnt = (((size_t)a.nq * 4) + 15) & ~(size_t)15;
    const size_t b_tau = (size_t)nqt * QT_ROWS * 4;
    const uint64_t n_chunks = filtered ? a.n_chunks : 0;
    const size_t b_split = bf16 ? (size_t)nqt * QT_ROWS * 1024 : 0;
    const size_t o_split = (b_cand + 2 * b_cnt + 3 * b_tau + 255) & ~(size_t)255;  // tau | thr | rq
    const size_t b_head = o_split + b_split;
    int rc = ensure_scratch(ctx, b_head + (size_t)n_chunks * sizeof(uint64_t) + 64);
    if (rc) return rc;
    char *base = reinterpret_cast<char *>(ctx->d_scratch);
    key_t64 *cand = reinterpret_cast<key_t64 *>(base);
    unsigned int *counts = reinterpret_cast<unsigned int *>(base + b_cand);
    unsigned int *overflow = reinterpret_cast<unsigned int *>(base + b_cand + b_cnt);
    float *tau = reinterpret_cast<float *>(base + b_cand + 2 * b_cnt);
    float *qconst = tau + (size_t)nqt * QT_ROWS;   // [nqt*32][2]
    uint64_t *chunk_table = reinterpret_cast<uint64_t *>(base + b_head);
    uint32_t *q_split = reinterpret_cast<uint32_t *>(base + o_split);
    if (f16x1)
        hipLaunchKernelGGL(split_queries_f16x1_kernel, dim3(nqt * QT_ROWS / 4), dim3(256), 0, ctx->stream, a.queries, a.nq,
                           nqt * QT_ROWS, q_split);
    else if (f16x2)
        hipLaunchKernelGGL(split_queries_f16_kernel, dim3(nqt * QT_ROWS / 4), dim3(256), 0, ctx->stream, a.queries, a.nq,
                           nqt * QT_ROWS, q_split);
    else if (bf16)
        hipLaunchKernelGGL(split_queries_kernel, dim3(nqt * QT_ROWS / 2), dim3(256), 0, ctx->stream, a.queries, a.nq,
                           nqt * QT_ROWS, q_split);
    if (rowreg)
        hipLaunchKernelGGL(query_consts_kernel, dim3(nqt * QT_ROWS / 4), dim3(256), 0, ctx->stream, a.queries, a.nq,
                           nqt * QT_ROWS, qconst, (f16x2 || f16x1) ? 1 : 0);
    if (filtered && (rc = launch_build_chunk_table(ctx, a.ranges, a.range_chunk_prefix, a.n_ranges, n_chunks, chunk_table))) return rc;
    SMT_HIP_CHECK(hipMemsetAsync(counts, 0, 2 * b_cnt, ctx->stream));
    hipLaunchKernelGGL(fill_f32_kernel, dim3((nqt * QT_ROWS + 255) / 256), dim3(256), 0, ctx->stream, tau,
                       __builtin_inff(), nqt * QT_ROWS);

    // level plan: strides ratio^(L-1) ... ratio, 1 with level 0 <= LEVEL0_MAX_TILES tiles
    const int LEVEL_RATIO = kp <= LEVEL_RATIO_KP_LIMIT ? LEVEL_RATIO_SMALL_K : LEVEL_RATIO_LARGE_K;
    const uint64_t n_tiles = filtered ? (n_chunks + 7) / 8 : (a.rows + 31) / 32;  // filtered: a tile = 8 chunks of <= 4 rows
    int L = 1;
    uint64_t s0 = 1;
    while ((n_tiles + s0 - 1) / s0 > (uint64_t)LEVEL0_MAX_TILES) { s0 *= LEVEL_RATIO; ++L; }
    int blocks = ctx->tune.gemm_blocks > 0 ? ctx->tune.gemm_blocks : ctx->num_cus;

    uint64_t stride = s0;
    for (int lev = 0; lev < L; ++lev, stride /= LEVEL_RATIO) {
        const uint64_t multiples = (n_tiles + stride - 1) / stride;                        // u in [0, multiples)
        const uint64_t parents = lev == 0 ? 0 : (multiples + LEVEL_RATIO - 1) / LEVEL_RATIO;  // u % 16 == 0
        GemmParams g;
        g.corpus = a.corpus;
        g.n_rows = a.rows;
        g.queries = a.queries;
        g.queries_split = bf16 ? q_split : nullptr;
        g.nq = a.nq;
        g.nqt = nqt;
        g.level_tiles = multiples - parents;
        g.tile_begin = 0;
        g.stride = stride;
        g.skip16 = lev == 0 ? 0 : LEVEL_RATIO;
        g.qsplit = 1;
        g.tau = tau;
        g.qconst = qconst;
        g.cand = cand;
        g.counts = counts;
        g.chunk_table = filtered ? chunk_table : nullptr;
        g.stamps = reinterpret_cast<unsigned long long *>(ctx->tune.scan_debug_ptr);
        g.buffered = lev > 0 && ctx->tune.gemm_buffered != 0;
        g.image = use_image ? a.image : nullptr;
        g.image_zero = use_image ? a.image_zero : nullptr;
        g.n_chunks = n_chunks;
        if (g.level_tiles > 0 && rowreg) {
            const uint64_t need_blocks = (g.level_tiles + RR_WAVES - 1) / RR_WAVES;
            int nb = (int)std::min<uint64_t>((uint64_t)blocks, need_blocks);
            if (ctx->tune.gemm_qsplit && need_blocks < (uint64_t)blocks) {   // small levels: split the query tiles over more blocks
                g.qsplit = (uint32_t)std::min<uint64_t>(nqt, std::max<uint64_t>(1, (uint64_t)2 * blocks / need_blocks));
                nb = (int)(need_blocks * g.qsplit);
            }
            // THE LAST LEVEL IN TWO PARTS when it is MFMA-bound (a streamed sweep): its thresholds come from a 1/ratio sample and
            // admit ~ratio x k' rows per query -- 12 % of the (row tile, query tile) products of a 1000 x 10 M batch nominate
            // something, and a nominating wave holds its block's other seven at the ring barrier (wave timeline: 63 % of the
            // barriers had such a straggler).  After the first eighth of the level a select pass tightens the thresholds to what
            // 18 % of the rows know (~3 x fewer nominations for the remaining 7/8); it costs one more launch and select pass.
            // the LDS nomination buffer pays where a wave's sweep nominates fewer pairs than it holds (a level admits ~ratio x k'
            // rows per query): the thin early levels go straight to the lists, one slot grab per lane and tile -- fewer atomics
            // on the same thousand counters, which is what those levels are bound by
            if ((double)LEVEL_RATIO * kp * a.nq / ((double)g.level_tiles * g.qsplit) > 0.75 * RR_CB_CAP) g.buffered = 0;
            const uint64_t level_end = g.level_tiles;
            uint64_t part_end = level_end;
            if (ctx->tune.gemm_split_last && lev == L - 1 && lev > 0 && nqt > (uint32_t)(f16x1 ? RrGeom<2>::SLOTS : RR_SLOTS) &&
                level_end >= (uint64_t)64 * blocks * RR_WAVES)
                part_end = (level_end / 8 + (uint64_t)blocks * RR_WAVES - 1) / ((uint64_t)blocks * RR_WAVES) * ((uint64_t)blocks * RR_WAVES);
            for (;;) {
                g.level_tiles = part_end;
                prof_begin(ctx, "gemm");
                if (f16x1 && use_image) hipLaunchKernelGGL((gemm_rowreg_kernel<2, true>), dim3(nb), dim3(RR_THREADS), (size_t)RrGeom<2>::SMEM, ctx->stream, g);
                else if (f16x2 && use_image) hipLaunchKernelGGL((gemm_rowreg_kernel<1, true>), dim3(nb), dim3(RR_THREADS), (size_t)RrGeom<1>::SMEM, ctx->stream, g);
                else if (f16x1) hipLaunchKernelGGL(gemm_rowreg_kernel<2>, dim3(nb), dim3(RR_THREADS), (size_t)RrGeom<2>::SMEM, ctx->stream, g);
                else if (f16x2) hipLaunchKernelGGL(gemm_rowreg_kernel<1>, dim3(nb), dim3(RR_THREADS), (size_t)RrGeom<1>::SMEM, ctx->stream, g);
                else hipLaunchKernelGGL(gemm_rowreg_kernel<0>, dim3(nb), dim3(RR_THREADS), (size_t)RrGeom<0>::SMEM, ctx->stream, g);
                prof_end(ctx, "gemm");
                if (part_end == level_end) break;
                LevelSelectParams mid;
                mid.cand = cand;
                mid.counts = counts;
                mid.tau = tau;
                mid.overflow = overflow;
                mid.kp = kp;
                mid.qconst = qconst;
                prof_begin(ctx, "select");
                hipLaunchKernelGGL(level_select_kernel, dim3(a.nq), dim3(1024), 0, ctx->stream, mid);
                prof_end(ctx, "select");
                g.tile_begin = part_end;
                part_end = level_end;
            }
        } else if (g.level_tiles > 0 && lds_rows) {
            const uint64_t need_blocks = (g.level_tiles + LR_WAVES - 1) / LR_WAVES;
            const int nb = (int)std::min<uint64_t>((uint64_t)blocks, need_blocks);
            const size_t smem = nqt <= 1 ? (size_t)LrGeom<1>::SMEM : (size_t)LrGeom<2>::SMEM;
            prof_begin(ctx, "gemm");
            const bool nt = ctx->tune.gemm_dma_nt != 0;
            if (bf16) lr_dispatch<true>(ctx, nqt, filtered, nt, nb, smem, g);
            else lr_dispatch<false>(ctx, nqt, filtered, nt, nb, smem, g);
            prof_end(ctx, "gemm");
        } else if (g.level_tiles > 0) {
            const uint64_t need_blocks = (g.level_tiles + GEMM_WAVES - 1) / GEMM_WAVES;
            int nb = (int)std::min<uint64_t>((uint64_t)blocks, need_blocks);
            // a level with fewer row-tile groups than CUs is a latency-bound sweep over the query tiles:
            // spread the query tiles over about two blocks per CU
            if (ctx->tune.gemm_qsplit && need_blocks < (uint64_t)blocks) {
                g.qsplit = (uint32_t)std::min<uint64_t>(nqt, std::max<uint64_t>(1, (uint64_t)2 * blocks / need_blocks));
                nb = (int)(need_blocks * g.qsplit);
            }
            prof_begin(ctx, "gemm");
            if (bf16) hipLaunchKernelGGL(gemm_level_kernel<true>, dim3(nb), dim3(GEMM_THREADS), gemm_smem_bytes(nqt), ctx->stream, g);
            else hipLaunchKernelGGL(gemm_level_kernel<false>, dim3(nb), dim3(GEMM_THREADS), gemm_smem_bytes(nqt), ctx->stream, g);
            prof_end(ctx, "gemm");
        }
        LevelSelectParams ls;
        ls.cand = cand;
        ls.counts = counts;
        ls.tau = tau;
        ls.overflow = overflow;
        ls.kp = kp;
        ls.qconst = rowreg ? qconst : nullptr;
        prof_begin(ctx, "select");
        hipLaunchKernelGGL(level_select_kernel, dim3(a.nq), dim3(1024), 0, ctx->stream, ls);
        prof_end(ctx, "select");
    }
    SMT_HIP_CHECK(hipGetLastError());

    // each query now has ONE sorted list of kp keys at the head of its buffer
    SelectArgs sel;
    sel.corpus = a.corpus;
    sel.queries = a.queries;
    sel.nq = a.nq;
    sel.lists = cand;
    sel.n_lists = 1;
    sel.kp = kp;
    sel.list_stride = CAND_CAP;
    sel.k_out = a.k_out;
    sel.ws_threshold = a.ws_threshold;
    sel.ws_thr_score = a.ws_thr_score;
    sel.row_base = a.row_base;
    sel.out_rows = a.out_rows;
    sel.out_dist = a.out_dist;
    sel.out_counts = a.out_counts;
    sel.out_stride = a.out_stride;
    sel.f32_err = f16x1 ? F32_ERR_F16X1 : f16x2 ? F32_ERR_F16X2 : bf16 ? F32_ERR_BF16X3 : F32_ERR_MFMA;
    sel.out_uncertain = a.out_uncertain;
    rc = launch_select(ctx, sel);
    if (rc) return rc;

    // overflow check (host sync: a batch is tens of milliseconds, the flag read is noise)
    std::vector<unsigned int> h_over(a.nq);
    SMT_HIP_CHECK(hipMemcpyAsync(h_over.data(), overflow, (size_t)a.nq * 4, hipMemcpyDeviceToHost, ctx->stream));
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (uint32_t q = 0; q < a.nq; ++q) {
        if (!h_over[q]) continue;
        ScanArgs one = a;  // exact fallback for this query through K2
        one.queries = a.queries + (size_t)q * 256;
        one.nq = 1;
        one.out_rows = a.out_rows + (size_t)q * ostride;
        one.out_dist = a.out_dist + (size_t)q * ostride;
        one.out_counts = a.out_counts ? a.out_counts + q : nullptr;
        one.out_uncertain = a.out_uncertain ? a.out_uncertain + q : nullptr;
        if ((rc = launch_scan_topk(ctx, one))) return rc;
    }
    return SMT_OK;
}

__global__ void set_qconst_thresholds_kernel(float *qconst, const float *tau, uint32_t nq)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) qconst[2 * q] = score_threshold(tau[q], qconst[2 * q + 1]);
}

// split image of n rows of 256 f32 (queries, k-means centroids) for the bf16 x 3 kernels of other translation units
int launch_split_rows_bf16(smt_ctx *ctx, const float *rows, uint32_t n, uint32_t n_pad, uint32_t *out)
{
    hipLaunchKernelGGL(split_queries_kernel, dim3((n_pad + 1) / 2), dim3(256), 0, ctx->stream, rows, n, n_pad, out);
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

// ---- batched threshold pass: the exhaustive re-answer of MANY uncertain queries in one sweep (api.cpp).  Every row
// whose nominating distance is <= tau[q] lands in query q's candidate buffer: gemm_rowreg_kernel over all tiles as a
// single level with preset thresholds.  A query with more than CAND_CAP such rows reports count > CAND_CAP (the caller
// re-answers it with the streaming K4 scan).  Buffers live in the context's scratch until the next launch.
int launch_gemm_threshold(smt_ctx *ctx, const float *corpus, uint64_t rows, const float *queries, uint32_t nq,
                          const float *tau, const key_t64 **cand_out, const unsigned int **counts_out, uint32_t *cand_stride)
{
    SMT_REQUIRE(nq >= 1 && rows >= 1 && rows < 0xFFFFFFFFull, "threshold pass: bad sizes");
    if (int rc_attr = ensure_gemm_attrs(ctx)) return rc_attr;
    const uint32_t nqt = (nq + QT_ROWS - 1) / QT_ROWS;
    const size_t b_cand = (size_t)nq * CAND_CAP * sizeof(key_t64);
    const size_t b_cnt = (((size_t)nq * 4) + 255) & ~(size_t)255;
    const size_t b_qc = (((size_t)nqt * QT_ROWS * 8) + 255) & ~(size_t)255;
    const size_t b_split = (size_t)nqt * QT_ROWS * 1024;
    int rc = ensure_scratch(ctx, b_cand + b_cnt + b_qc + b_split + 256);
    if (rc) return rc;
    char *base = reinterpret_cast<char *>(ctx->d_scratch);
    key_t64 *cand = reinterpret_cast<key_t64 *>(base);
    unsigned int *counts = reinterpret_cast<unsigned int *>(base + b_cand);
    float *qconst = reinterpret_cast<float *>(base + b_cand + b_cnt);
    uint32_t *q_split = reinterpret_cast<uint32_t *>(base + b_cand + b_cnt + b_qc);
    SMT_HIP_CHECK(hipMemsetAsync(counts, 0, b_cnt, ctx->stream));
    hipLaunchKernelGGL(split_queries_kernel, dim3(nqt * QT_ROWS / 2), dim3(256), 0, ctx->stream, queries, nq, nqt * QT_ROWS, q_split);
    hipLaunchKernelGGL(query_consts_kernel, dim3(nqt * QT_ROWS / 4), dim3(256), 0, ctx->stream, queries, nq, nqt * QT_ROWS, qconst, 0);
    hipLaunchKernelGGL(set_qconst_thresholds_kernel, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, qconst, tau, nq);
    GemmParams g;
    g.corpus = corpus;
    g.n_rows = rows;
    g.queries = queries;
    g.queries_split = q_split;
    g.nq = nq;
    g.nqt = nqt;
    g.level_tiles = (rows + 31) / 32;
    g.stride = 1;
    g.skip16 = 0;
    g.qsplit = 1;
    g.tau = nullptr;
    g.qconst = qconst;
    g.cand = cand;
    g.counts = counts;
    g.chunk_table = nullptr;
    g.stamps = nullptr;
    g.buffered = 0;
    g.image = nullptr;
    g.image_zero = nullptr;
    g.tile_begin = 0;
    g.n_chunks = 0;
    const int blocks = ctx->tune.gemm_blocks > 0 ? ctx->tune.gemm_blocks : ctx->num_cus;
    const int nb = (int)std::min<uint64_t>((uint64_t)blocks, (g.level_tiles + RR_WAVES - 1) / RR_WAVES);
    prof_begin(ctx, "gemm_thr");   // (always bf16 x 3: the tighter band collects fewer rows)
    hipLaunchKernelGGL(gemm_rowreg_kernel<false>, dim3(nb), dim3(RR_THREADS), (size_t)RR_SMEM, ctx->stream, g);
    prof_end(ctx, "gemm_thr");
    SMT_HIP_CHECK(hipGetLastError());
    *cand_out = cand;
    *counts_out = counts;
    *cand_stride = CAND_CAP;
    return SMT_OK;
}

int launch_pack_image(smt_ctx *ctx, const float *corpus, uint64_t n_rows, uint64_t first_tile, uint64_t n_tiles, void *image,
                      uint32_t *image_zero)
{
    if (n_tiles == 0) return SMT_OK;
    prof_begin(ctx, "pack_image");
    hipLaunchKernelGGL(pack_image_kernel, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, ctx->stream, corpus, n_rows, first_tile,
                       n_tiles, reinterpret_cast<uint32_t *>(image), image_zero);
    prof_end(ctx, "pack_image");
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

}  // namespace smt
