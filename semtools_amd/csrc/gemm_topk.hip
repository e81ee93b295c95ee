// gemm_topk.hip -- K3: batched queries (the level plan, the per-batch helper kernels, launch_gemm_topk; the three MFMA kernels
// live in gemm_rowreg.hip, gemm_ldsrow.hip and gemm_level.hip, what they share in gemm.h).  S = C x Q^T on the MFMA pipes with the top-k candidate selection fused
// into the epilogue -- the nq x N score matrix (40 GB at 1k x 10M) is never materialised.  The scores only NOMINATE
// candidates (k + guard per query); answers are re-scored exactly in f64 and proved complete by the select stage.
//
// The kernels, and who runs when (launch_gemm_topk):
//   gemm_rowreg_kernel<F16X2>   DEFAULT for every unfiltered batch.  Row tiles arrive with coalesced loads, are split
//                               into 16-bit operands once and transposed through LDS; bf16 x 3 products below 128
//                               queries, f16 x 2 from there (tuning key gemm_nominate).  DESIGN.md 4.3.
//   gemm_ldsrow_kernel<..>      range-filtered batches (chunk table), <= 64 queries per pass; f32 or bf16 x 3 MFMAs.
//   gemm_level_kernel<BF16>     the round-1 corpus-stationary kernel: f32 MFMAs (v_mfma_f32_32x32x2_f32, exact f32,
//                               157 TF peak) when gemm_bf16x3 = 0, or bf16 x 3 when gemm_rowreg = 0.  The level scheme,
//                               the epilogue and the candidate buffers described below are shared by all three.
//   level_select_kernel, split_queries_*_kernel, query_consts_kernel: per-level / per-batch helpers.
//   launch_gemm_threshold       one sweep with preset thresholds: the batched exhaustive re-answer (search.cpp).
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
#include "gemm.h"

namespace smt {


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

// What the head of a row-register batch sets up per query besides the split image: (score threshold, 1/|q|) -- for the fp16 modes the
// operands are unit vectors times 2^10 and 2^8, so 1/|q| becomes the inverse of that scale -- and the resets (nomination count and
// overflow flag 0, tau +inf).  Riding in the split kernel it is ONE launch at the head of a call instead of four: the GPU is idle
// there and waits out every launch latency (~10 us apiece of a 0.9 ms call).
struct BatchHead {
    float *qconst;          // [nq_pad][2], or nullptr: nothing but the image
    unsigned int *counts, *overflow;
    float *tau;
    float *fill = nullptr;  // the bootstrap level's slot minima: n_fill floats preset to +inf by the same launch
    uint32_t n_fill = 0;
};
__device__ __forceinline__ void batch_head_fill(const BatchHead &b)
{
    if (b.fill == nullptr) return;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < b.n_fill; i += stride) b.fill[i] = __builtin_inff();
}
__device__ __forceinline__ void batch_head_f16(const BatchHead &b, uint32_t q, uint32_t nq, float a2, int lane)
{
    if (b.qconst && lane == 0) {
        const float rq = a2 == 0.0f ? 0.0f : F16X2_INV_SCALE;
        b.qconst[2 * q] = score_threshold(q < nq ? __builtin_inff() : -1.0f, q < nq ? rq : 0.0f);  // padding: zero query, tau < 0
        b.qconst[2 * q + 1] = q < nq ? rq : 0.0f;
        b.tau[q] = __builtin_inff();
        if (q < nq) { b.counts[q] = 0; b.overflow[q] = 0; }
    }
}

// The f16 x 2 image of the queries (same row layout: K-step m, half h -> 16 B of hi, 16 B of lo): the UNIT query times
// 2^8, split into two fp16 parts.  One wave per query (the norm is needed first).
__global__ void split_queries_f16_kernel(const float *queries, uint32_t nq, uint32_t nq_pad, uint32_t *out, BatchHead head = {})
{
    batch_head_fill(head);
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
    batch_head_f16(head, q, nq, a2, lane);
}

// The f16 x 1 image: the hi halves only, 512 B per query -- (K-step m, half h) -> 16 B at word 8 m + 4 h.
__global__ void split_queries_f16x1_kernel(const float *queries, uint32_t nq, uint32_t nq_pad, uint32_t *out, BatchHead head = {})
{
    batch_head_fill(head);
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
    batch_head_f16(head, q, nq, a2, lane);
}

// (also the per-batch resets when `counts` is given: counts / overflow flags to 0, tau to +inf -- one launch instead of three at
// the head of a call whose GPU is idle and waits out every launch latency: ~20 us of a 0.9 ms call)
__global__ void query_consts_kernel(const float *queries, uint32_t nq, uint32_t nq_pad, float *qconst, int f16x2,
                                    unsigned int *counts = nullptr, unsigned int *overflow = nullptr, float *tau = nullptr,
                                    float *fill = nullptr, uint32_t n_fill = 0)
{
    {
        BatchHead f{};
        f.fill = fill;
        f.n_fill = n_fill;
        batch_head_fill(f);
    }
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
        if (counts) {
            tau[q] = __builtin_inff();
            if (q < nq) { counts[q] = 0; overflow[q] = 0; }
        }
    }
}

// Per query: keep the kp best of the candidates gathered so far (sorted, at the head of the buffer), publish tau = kp-th
// distance, flag overflow.
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

// The first threshold of a row-register batch: per query the kp-th smallest of the bootstrap level's slot minima (GemmParams::
// tile_min, [<= 1024 slots][nq_pad]; a slot = the minimum over the tiles that share it).  The tiles are distinct, so at least kp ROWS
// have a nominating distance <= that value: it bounds the final kp-th distance from above, which is all a level threshold has to do.
// One WAVE per pair of queries (32 lanes each, 32 slot values per lane in registers, read straight from the [slot][query] matrix: 4-byte
// reads of 1024 different rows -- 16 x over-fetch of an L2-resident 4 MB, latency-bound at ~3 us); the k'-th smallest is then built
// bit by bit, per bit 32 ballots and their popcounts.  Earlier forms, all one 1024-thread block per query TILE: a bitonic sort of the
// 32 LDS rows (150 us -- more than the three levels it replaced), dependent loads + shuffle sums (35 us), one round of loads + ballots
// (35 us still: sixteen waves on the four SIMDs of one CU issue 4 x 31 x 32 compares each, and at most 32 CUs were busy at all).
__global__ void __launch_bounds__(64) bootstrap_tau_kernel(const float *tile_min, uint32_t n_slots, uint32_t nq, uint32_t nq_pad, uint32_t kp,
                                                           float *tau, float *qconst)
{
    const uint32_t ql = threadIdx.x >> 5, l = threadIdx.x & 31;
    const uint32_t q = blockIdx.x * 2 + ql;
    if (blockIdx.x * 2 >= nq) return;           // (both queries of the wave are padding)
    const bool upper = ql != 0;
    uint32_t v[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        const uint32_t t = l + 32u * e;
        v[e] = t < n_slots ? __float_as_uint(tile_min[(size_t)t * nq_pad + q]) : 0x7F800000u;
    }
    // Each lane keeps only its KEEP smallest values (insertion through a min / max chain): the k'-th smallest of that SUBSET is >= the
    // k'-th smallest of all 1024 -- still at least k' slots lie at or below it, so it is still a valid threshold -- and equal to it
    // unless one lane holds more than KEEP of the k' best (k' = 34 over 32 lanes: ~1 per lane expected).  The bit loop then counts
    // KEEP values per lane instead of 32.
    constexpr int KEEP = 6;
    uint32_t s6[KEEP];
#pragma unroll
    for (int i = 0; i < KEEP; ++i) s6[i] = 0x7F800000u;
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        uint32_t x = v[e];
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            const uint32_t lo = x < s6[i] ? x : s6[i], hi = x < s6[i] ? s6[i] : x;
            s6[i] = lo;
            x = hi;
        }
    }
    uint32_t ans = 0;
    for (int bit = 30; bit >= 0; --bit) {
        const uint32_t test = ans | (1u << bit);
        uint32_t c_lo = 0, c_hi = 0;
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(s6[i] < test);
            c_lo += (uint32_t)__builtin_popcount((uint32_t)m);
            c_hi += (uint32_t)__builtin_popcount((uint32_t)(m >> 32));
        }
        if ((upper ? c_hi : c_lo) < kp) ans = test;   // fewer than kp values lie below `test`: the kp-th smallest is >= test
    }
    if (l == 0 && q < nq) {
        // (fewer than kp finite values in the subset: the construction ends at the pattern of +inf or above -- no threshold yet)
        const float t = kp <= n_slots && ans < 0x7F800000u ? __uint_as_float(ans) : __builtin_inff();
        tau[q] = t;
        qconst[2 * q] = score_threshold(t, qconst[2 * q + 1]);
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

static int ensure_gemm_attrs(smt_ctx *ctx)
{
    if (!(ctx->attr_done & ATTR_GEMM)) {  // per context == per device
        SMT_HIP_CHECK(gemm_level_set_attrs());
        SMT_HIP_CHECK(gemm_rowreg_set_attrs());
        SMT_HIP_CHECK(gemm_ldsrow_set_attrs());
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
    // ... and the range-filtered ones whose rows fill the aligned 32-row tiles they touch well enough (tiles_dense, common.h): the
    // kernel then walks a TILE TABLE (tile | row mask, build_tile_table_kernel) instead of all tiles -- the unit of the fp16 operand
    // image, so a workspace search over a subset of documents (src/workspace/store.rs:507-515) gets the image and the fp16 modes too.
    const bool rowreg = ctx->tune.gemm_bf16x3 && ctx->tune.gemm_rowreg &&
                        (!filtered || (a.range_tile_prefix != nullptr && tiles_dense(a.n_virtual, a.n_vtiles)));
    // How gemm_rowreg_kernel nominates (tuning key gemm_nominate: 0 auto, 1 bf16 x 3, 2 f16 x 2, 3 f16 x 1).  The fp16 modes
    // issue 2/3 resp. 1/3 of the MFMAs of bf16 x 3 -- what large batches are bound by -- for a wider certificate band
    // (5.2e-4 / 1.0e-3 against 7e-5): auto takes f16 x 2 from 128 queries and f16 x 1 from 129 (until round 5: 256) queries on shards of at most
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
    // (every automatic fp16 choice needs room for its guard band in the 64-entry lists -- k + 24 resp. k + 16 -- or most queries
    // would fail their certificate and be re-answered exhaustively: bf16 x 3 then)
    // (round 5, late, tools/probe_mid.py -> profiles/r05_nominate_modes.json, f32 rows, ms per call, f16 x 2 | f16 x 1: 10 M rows 160
    // queries 2.73 | 2.14, 200: 3.28 | 1.95, 224: 3.22 | 1.98; 2 M rows 200: 0.78 | 0.59; 128 queries: equal; no query re-answered --
    // five to seven query tiles are no longer HBM-bound with two MFMA passes: f16 x 1 from 129 queries, not 256)
    const bool f16x1 = rowreg && (ctx->tune.gemm_nominate == 3 ||
                                  (auto_fp16 && (nqt >= 5 || (have_image && nqt >= 4)) && a.k_out + 24 <= 64));
    const bool f16x2 = rowreg && !f16x1 && (ctx->tune.gemm_nominate == 2 || (auto_fp16 && (nqt >= 4 || have_image) && a.k_out + 16 <= 64));
    const bool use_image = have_image && (f16x1 || f16x2);
    // (Round 5 built the obvious alternative for ONE to FOUR queries over the image -- a single pass with per-wave candidate lists,
    // K2's structure over 512-byte rows on the fp16 MFMA pipe, no levels -- and measured it on one box against this plan, ms per
    // device-resident call: 1 query x 10 M rows 0.802 against 0.836, but 4 x 10 M 1.08 against 0.84, 1 x 1 M 0.148 against 0.132,
    // 2 x 2 M 0.310 against 0.228, one / three queries over a 5 M-row document subset 0.494 / 0.669 against 0.484 / 0.497
    // (profiles/r05_ab_image_scan.txt).  A wave's own k'-th is a loose threshold -- ~100 inserts per wave and query, each a
    // wave-wide affair out of MFMA accumulators -- where the levels' thresholds come from merged lists.  Not kept.)
    // guard band, see candidates_per_list (scan_kernels.hip): the wider the certificate band, the more rows are nominated
    // (the proof needs the k-th exact distance to lie 2 x the band below the worst nominated one): 8 / 16 / 24
    const uint32_t kp = std::min<uint32_t>(64, a.k_out + (uint32_t)std::max(ctx->tune.guard_band, f16x1 ? 24 : f16x2 ? 16 : 8));
    const bool lds_rows = !rowreg && ctx->tune.gemm_ldsrow && (filtered || nqt <= 2);
    if (filtered && !lds_rows && !rowreg) { set_error("range-filtered batches need the row-register or the LDS-row kernel (tuning keys gemm_rowreg, gemm_ldsrow)"); return SMT_E_UNSUPPORTED; }
    const uint32_t pass_nq = lds_rows ? 2 * QT_ROWS : GEMM_MAX_NQ;
    if (a.nq > pass_nq) {
        // (GEMM_MAX_NQ: the per-query thresholds of one gemm_level_kernel launch live in LDS beside the four
        // query-tile slots: larger batches are answered in chunks, each its own sweep over the corpus)
        SMT_REQUIRE(a.deliver == nullptr, "an answer delivered by the select kernel comes from ONE select launch (search.cpp keeps such calls below one pass)");
        for (uint32_t q0 = 0; q0 < a.nq; q0 += pass_nq) {
            ScanArgs c = a;
            c.nq = std::min<uint32_t>(pass_nq, a.nq - q0);
            c.queries = a.queries + (size_t)q0 * 256;
            c.out_rows = a.out_rows + (size_t)q0 * ostride;
            c.out_dist = a.out_dist + (size_t)q0 * ostride;
            c.out_counts = a.out_counts ? a.out_counts + q0 : nullptr;
            c.out_uncertain = a.out_uncertain ? a.out_uncertain + q0 : nullptr;
            c.out_status = a.out_status ? a.out_status + q0 : nullptr;
            const int rc_chunk = launch_gemm_topk(ctx, c);
            if (rc_chunk) return rc_chunk;
        }
        return SMT_OK;
    }
    if (gemm_level_smem_bytes(nqt) > 160 * 1024) { set_error("batch too large for one launch"); return SMT_E_UNSUPPORTED; }

    if (int rc_attr = ensure_gemm_attrs(ctx)) return rc_attr;

    // scratch: cand [nq][CAP] keys | counts [nq] | overflow [nq] | tau [nqt*32] | split queries [nqt*32][1 KiB] | chunk table
    const bool bf16 = ctx->tune.gemm_bf16x3 != 0;
    const size_t b_cand = (size_t)a.nq * CAND_CAP * sizeof(key_t64);
    const size_t b_cnt = (((size_t)a.nq * 4) + 15) & ~(size_t)15;
    const size_t b_tau = (size_t)nqt * QT_ROWS * 4;
    const uint64_t n_chunks = !filtered ? 0 : rowreg ? a.n_vtiles : a.n_chunks;   // entries of the chunk resp. tile table
    const size_t b_split = bf16 ? (size_t)nqt * QT_ROWS * 1024 : 0;
    const size_t o_split = (b_cand + 2 * b_cnt + 3 * b_tau + 255) & ~(size_t)255;  // tau | thr | rq
    const size_t b_head = o_split + b_split;
    // the row-register kernel's bootstrap level: every s0-th tile, at most BOOTSTRAP_MAX_TILES of them (level plan below)
    const uint64_t plan_tiles = !filtered ? (a.rows + 31) / 32 : rowreg ? n_chunks : (n_chunks + 7) / 8;
    const bool bootstrap = rowreg && plan_tiles > (uint64_t)LEVEL0_MAX_TILES && ctx->tune.gemm_bootstrap != 0;
    // Bootstrap stride (see the level plan below): every tile up to 2048 tiles; every 16th up to 128 Ki tiles (4 M rows) -- ONE
    // appended level over all tiles follows; beyond that every 64th (x 16 until <= BOOTSTRAP_MAX_TILES tiles are left).
    uint64_t boot_stride = 1;
    if (bootstrap && plan_tiles > 2048) {
        // Graded since the end of round 5 (tools/ab_boot_stride.py, profiles/r05_boot_stride.txt; ms per call, 1000 queries, every 16th
        // tile | the stride below): 66 k rows 0.67 | 0.44, 131 k 0.53 | 0.41, 262 k 0.59 | 0.51, 524 k 0.75 | 0.69, 1 M 1.01 | 1.02; 256
        // queries 66 k 0.27 | 0.19, 262 k 0.30 | 0.21, 524 k 0.34 | 0.26 -- a 1/16 sample of a SMALL corpus leaves thresholds that admit
        // 16 k' rows per query into few tiles (4 nominations per product at 2050 tiles), and the bootstrap's own pass is cheap there.
        // (tuning key gemm_boot_fine = 0: every 16th tile up to 131 072 tiles, as in round 4)
        boot_stride = !ctx->tune.gemm_boot_fine ? (plan_tiles <= (uint64_t)131072 ? 16 : 64)
                      : plan_tiles <= 4096 ? 2 : plan_tiles <= 8192 ? 4 : plan_tiles <= 32768 ? 8 : plan_tiles <= (uint64_t)131072 ? 16 : 64;
        while ((plan_tiles + boot_stride - 1) / boot_stride > (uint64_t)BOOTSTRAP_MAX_TILES) boot_stride *= 16;
    }
    const uint64_t boot_tiles = bootstrap ? (plan_tiles + boot_stride - 1) / boot_stride : 0;
    const size_t o_table = (b_head + 255) & ~(size_t)255;
    const size_t o_tmin = (o_table + (size_t)n_chunks * sizeof(uint64_t) + 255) & ~(size_t)255;
    const uint64_t boot_slots = std::min<uint64_t>(boot_tiles, BOOTSTRAP_SLOTS);
    int rc = ensure_scratch(ctx, o_tmin + (size_t)boot_slots * nqt * QT_ROWS * sizeof(float) + 64);
    if (rc) return rc;
    char *base = reinterpret_cast<char *>(ctx->d_scratch);
    key_t64 *cand = reinterpret_cast<key_t64 *>(base);
    unsigned int *counts = reinterpret_cast<unsigned int *>(base + b_cand);
    unsigned int *overflow = reinterpret_cast<unsigned int *>(base + b_cand + b_cnt);
    float *tau = reinterpret_cast<float *>(base + b_cand + 2 * b_cnt);
    float *qconst = tau + (size_t)nqt * QT_ROWS;   // [nqt*32][2]
    uint64_t *chunk_table = reinterpret_cast<uint64_t *>(base + o_table);
    float *tile_min = reinterpret_cast<float *>(base + o_tmin);
    uint32_t *q_split = reinterpret_cast<uint32_t *>(base + o_split);
    BatchHead head{qconst, counts, overflow, tau};   // (fp16 modes are row-register modes)
    if (bootstrap) {   // the bootstrap level's slot minima are preset to +inf by the head launch (one launch fewer in front of the batch)
        head.fill = tile_min;
        head.n_fill = (uint32_t)(boot_slots * nqt * QT_ROWS);
    }
    if (f16x1)
        hipLaunchKernelGGL(split_queries_f16x1_kernel, dim3(nqt * QT_ROWS / 4), dim3(256), 0, ctx->stream, a.queries, a.nq,
                           nqt * QT_ROWS, q_split, head);
    else if (f16x2)
        hipLaunchKernelGGL(split_queries_f16_kernel, dim3(nqt * QT_ROWS / 4), dim3(256), 0, ctx->stream, a.queries, a.nq,
                           nqt * QT_ROWS, q_split, head);
    else if (bf16)
        hipLaunchKernelGGL(split_queries_kernel, dim3(nqt * QT_ROWS / 2), dim3(256), 0, ctx->stream, a.queries, a.nq,
                           nqt * QT_ROWS, q_split);
    if (rowreg && !(f16x1 || f16x2))   // bf16 x 3: its split kernel has no wave per query
        hipLaunchKernelGGL(query_consts_kernel, dim3(nqt * QT_ROWS / 4), dim3(256), 0, ctx->stream, a.queries, a.nq,
                           nqt * QT_ROWS, qconst, 0, counts, overflow, tau, head.fill, head.n_fill);
    // (the table of a KEPT range set -- search.cpp -- is built the first time a call needs it and then only read)
    const uint64_t *use_table = chunk_table;
    if (filtered && rowreg) {
        if ((rc = range_tile_table(ctx, a, chunk_table, &use_table))) return rc;
    } else if (filtered && (rc = range_chunk_table(ctx, a, chunk_table, &use_table))) return rc;
    if (!rowreg) {
        SMT_HIP_CHECK(hipMemsetAsync(counts, 0, 2 * b_cnt, ctx->stream));
        hipLaunchKernelGGL(fill_f32_kernel, dim3((nqt * QT_ROWS + 255) / 256), dim3(256), 0, ctx->stream, tau,
                           __builtin_inff(), nqt * QT_ROWS);
    }

    // LEVEL PLAN.  A level = every stride-th tile that no coarser level has visited (`skip` = the ratio to the coarser level: the
    // multiples of it are left out; 0 = none left out), appended under the thresholds the levels before it produced.
    //  * appended-levels plan (rounds 1-3; the f32 / LDS-row kernels, gemm_bootstrap = 0): strides 16^(L-1) ... 16, 1, the first level
    //    <= LEVEL0_MAX_TILES tiles appended without a threshold.
    //  * bootstrap plan (gemm_rowreg_kernel, round 4): a bootstrap level of tile minima gives the first thresholds; then
    //      <= 2048 tiles: bootstrap over ALL tiles, one appended level over all of them (twice the work of a corpus of <= 64 k rows);
    //      <= 128 Ki tiles: bootstrap over every 2nd / 4th / 8th / 16th (4 Ki / 8 Ki / 32 Ki / 128 Ki tiles), one appended level over ALL tiles;
    //      larger: bootstrap over every 64th (x 16 while more than BOOTSTRAP_MAX_TILES are left), appended levels at strides
    //      ... 64, 4 -- the first visits every multiple of its stride, the bootstrap's tiles included -- and a LAST level of
    //      ratio 4: the 3/4 of the corpus it holds are appended under thresholds a quarter of the corpus produced (~4 k' rows per
    //      query), and the level before it is a quarter of the corpus -- no level is thin any more.  (Rounds 2-3 had a 1/16 level
    //      whose ~16 k' admissions per query fell on 6 % of the products -- 0.87 nominations per product against 0.06 in the main
    //      level -- and a main level split in two to tighten its thresholds after an eighth: profiles/r04_k3/.)
    const int LEVEL_RATIO = kp <= LEVEL_RATIO_KP_LIMIT ? LEVEL_RATIO_SMALL_K : LEVEL_RATIO_LARGE_K;
    const uint64_t n_tiles = plan_tiles;
    struct PlanLevel { uint64_t stride; int skip; };
    std::vector<PlanLevel> plan;
    int blocks = ctx->tune.gemm_blocks > 0 ? ctx->tune.gemm_blocks : ctx->num_cus;
    if (bootstrap) {
        // ---- BOOTSTRAP: every boot_stride-th tile, tile minima only, then the first thresholds; the appended levels start at
        // stride boot_stride / ratio (or 1) and their first one visits EVERY multiple of its stride, the bootstrap's tiles included
        // (1/ratio of that level: the price of a level without candidate lists)
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.corpus = a.corpus;
        g.n_rows = a.rows;
        g.queries = a.queries;
        g.queries_split = q_split;
        g.nq = a.nq;
        g.nqt = nqt;
        g.level_tiles = boot_tiles;
        g.tile_begin = 0;
        g.stride = boot_stride;
        g.skip16 = 0;
        g.qsplit = 1;
        g.tau = tau;
        g.qconst = qconst;
        g.cand = cand;
        g.counts = counts;
        g.tile_table = filtered ? use_table : nullptr;
        g.tile_min = tile_min;
        g.image = use_image ? a.image : nullptr;
        g.image_zero = use_image ? a.image_zero : nullptr;
        const uint64_t need_blocks = (boot_tiles + RR_WAVES - 1) / RR_WAVES;
        int nb = (int)std::min<uint64_t>((uint64_t)blocks, need_blocks);
        if (ctx->tune.gemm_qsplit && need_blocks < (uint64_t)blocks) {
            g.qsplit = (uint32_t)std::min<uint64_t>(nqt, std::max<uint64_t>(1, (uint64_t)2 * blocks / need_blocks));
            nb = (int)(need_blocks * g.qsplit);
        }
        prof_begin(ctx, "gemm");
        gemm_rowreg_launch(ctx, f16x1 ? 2 : f16x2 ? 1 : 0, use_image, nb, g);
        prof_end(ctx, "gemm");
        prof_begin(ctx, "select");
        hipLaunchKernelGGL(bootstrap_tau_kernel, dim3(nqt * QT_ROWS / 2), dim3(64), 0, ctx->stream, tile_min, (uint32_t)boot_slots, a.nq,
                           nqt * QT_ROWS, kp, tau, qconst);
        prof_end(ctx, "select");
        if (boot_stride <= 16) plan.push_back({1, 0});
        else {
            uint64_t st = boot_stride / 16;           // 4 x 16^j
            plan.push_back({st, 0});
            while (st > 4) { st /= 16; plan.push_back({st, 16}); }
            plan.push_back({1, 4});
        }
    } else {
        uint64_t s0 = 1;
        while ((n_tiles + s0 - 1) / s0 > (uint64_t)LEVEL0_MAX_TILES) s0 *= LEVEL_RATIO;
        plan.push_back({s0, 0});
        for (uint64_t st = s0; st > 1;) { st /= LEVEL_RATIO; plan.push_back({st, LEVEL_RATIO}); }
    }
    const int L = (int)plan.size();

    for (int lev = 0; lev < L; ++lev) {
        const uint64_t stride = plan[lev].stride;
        const int skip = plan[lev].skip;
        const uint64_t multiples = (n_tiles + stride - 1) / stride;                        // u in [0, multiples)
        const uint64_t parents = skip == 0 ? 0 : (multiples + (uint64_t)skip - 1) / (uint64_t)skip;   // u % skip == 0: visited before
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
        g.skip16 = skip;
        g.qsplit = 1;
        g.tau = tau;
        g.qconst = qconst;
        g.cand = cand;
        g.counts = counts;
        g.chunk_table = filtered && !rowreg ? use_table : nullptr;
        g.tile_table = filtered && rowreg ? use_table : nullptr;
        g.tile_min = nullptr;
        g.stamps = reinterpret_cast<unsigned long long *>(ctx->tune.scan_debug_ptr);
        g.buffered = (lev > 0 || bootstrap) && ctx->tune.gemm_buffered != 0;   // (thresholds exist: nominations are few)
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
            const int admit_ratio = skip ? skip : 16;   // rows admitted per query ~ admit_ratio x k'
            if ((double)admit_ratio * kp * a.nq / ((double)g.level_tiles * g.qsplit) > 0.75 * RR_CB_CAP) g.buffered = 0;
            const uint64_t level_end = g.level_tiles;
            uint64_t part_end = level_end;
            // ... and the FIRST appended level of a bootstrap plan (the quarter-corpus level) likewise: its thresholds come from the
            // bootstrap's 1/64 sample (~16 k' admissions per query, 0.2 nominations per product); after its first quarter a select
            // pass tightens them to what 1/16 of the corpus knows.  Timelines on one box (1000 x 10 M, image): 1598 -> 431 + 1061 us
            // for the level and 42 -> 11 + 9 us of select: 5.82 -> 5.69 ms per call (-2.3 %); from f32 rows 6.55 -> 6.42; 512
            // queries 3.27 -> 3.16.  The same for the ONE appended level of a mid-sized corpus (<= 128 Ki tiles: bootstrap over every
            // 16th tile, then everything under ~16 k' admissions per query): 1000 x 1 M 867 -> 803 us, x 2 M 1441 -> 1347, x 4 M
            // 2513 -> 2428, 512 x 1 M 512 -> 482.  (Tuning key gemm_split_last: 0 none, 1 a ratio-16 last level only, 2 both.)
            const bool first_after_boot = ctx->tune.gemm_split_last >= 2 && bootstrap && lev == 0;   // (also when it is the only level)
            const bool split_level = lev == L - 1 || first_after_boot;
            if (ctx->tune.gemm_split_last && split_level && (lev > 0 || bootstrap) && admit_ratio >= 16 && nqt > (uint32_t)(f16x1 ? RrGeom<2>::SLOTS : RR_SLOTS) &&
                level_end >= (uint64_t)(first_after_boot ? 12 : 64) * blocks * RR_WAVES)
                part_end = (level_end / (first_after_boot ? 4 : 8) + (uint64_t)blocks * RR_WAVES - 1) / ((uint64_t)blocks * RR_WAVES) * ((uint64_t)blocks * RR_WAVES);
            for (;;) {
                g.level_tiles = part_end;
                prof_begin(ctx, "gemm");
                gemm_rowreg_launch(ctx, f16x1 ? 2 : f16x2 ? 1 : 0, use_image, nb, g);
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
            prof_begin(ctx, "gemm");
            gemm_ldsrow_launch(ctx, bf16, nqt, filtered, ctx->tune.gemm_dma_nt != 0, nb, g);
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
            gemm_level_launch(ctx, bf16, nqt, nb, g);
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
    sel.out_status = a.out_status;
    sel.overflow = overflow;
    sel.deliver = a.deliver;
    rc = launch_select(ctx, sel);
    if (rc) return rc;

    return SMT_OK;   // (nothing synchronises: a query whose buffer overflowed is flagged by the select -- SelectArgs::overflow)
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

// ---- batched threshold pass: the exhaustive re-answer of MANY uncertain queries in one sweep (search.cpp).  Every row
// whose nominating distance is <= tau[q] lands in query q's candidate buffer: gemm_rowreg_kernel over all tiles as a
// single level with preset thresholds.  A query with more than CAND_CAP such rows reports count > CAND_CAP (the caller
// re-answers it with the streaming K4 scan).  Buffers live in the context's scratch until the next launch.
int launch_gemm_threshold(smt_ctx *ctx, const float *corpus, uint64_t rows, const void *image, const uint32_t *image_zero,
                          const float *queries, uint32_t nq, const float *tau, const key_t64 **cand_out,
                          const unsigned int **counts_out, uint32_t *cand_stride)
{
    const bool f16 = image != nullptr;   // over the corpus' operand image: f16 x 2 (the caller widened tau by F32_ERR_F16X2)
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
    if (f16) hipLaunchKernelGGL(split_queries_f16_kernel, dim3(nqt * QT_ROWS / 4), dim3(256), 0, ctx->stream, queries, nq, nqt * QT_ROWS, q_split);
    else hipLaunchKernelGGL(split_queries_kernel, dim3(nqt * QT_ROWS / 2), dim3(256), 0, ctx->stream, queries, nq, nqt * QT_ROWS, q_split);
    hipLaunchKernelGGL(query_consts_kernel, dim3(nqt * QT_ROWS / 4), dim3(256), 0, ctx->stream, queries, nq, nqt * QT_ROWS, qconst, f16 ? 1 : 0);
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
    g.tile_table = nullptr;
    g.stamps = nullptr;
    g.buffered = 0;
    g.image = image;
    g.image_zero = image_zero;
    g.tile_begin = 0;
    g.n_chunks = 0;
    const int blocks = ctx->tune.gemm_blocks > 0 ? ctx->tune.gemm_blocks : ctx->num_cus;
    const int nb = (int)std::min<uint64_t>((uint64_t)blocks, (g.level_tiles + RR_WAVES - 1) / RR_WAVES);
    // from the f32 rows: bf16 x 3 (the tightest band collects the fewest rows); with the operand image: f16 x 2 over half the bytes
    prof_begin(ctx, "gemm_thr");
    gemm_rowreg_launch(ctx, f16 ? 1 : 0, f16, nb, g);
    prof_end(ctx, "gemm_thr");
    SMT_HIP_CHECK(hipGetLastError());
    *cand_out = cand;
    *counts_out = counts;
    *cand_stride = CAND_CAP;
    return SMT_OK;
}

}  // namespace smt
