// gemm_level.hip -- gemm_level_kernel: the round-1 corpus-stationary K3 kernel, f32 MFMAs (tuning key gemm_bf16x3 = 0: the
// "f32 MFMA Q x C^T" form BASELINE config c3 names) or bf16 x 3 (gemm_rowreg = 0).  Design notes: gemm_topk.hip, DESIGN.md 4.3.
#include "gemm.h"

namespace smt {

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

size_t gemm_level_smem_bytes(uint32_t nqt)
{
    return (size_t)4 * QT_F4 * 16 + (size_t)nqt * QT_ROWS * 4 * 2 + 64;
}

hipError_t gemm_level_set_attrs()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_level_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_level_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

void gemm_level_launch(smt_ctx *ctx, bool bf16, uint32_t nqt, int nb, const GemmParams &g)
{
    if (bf16) hipLaunchKernelGGL(gemm_level_kernel<true>, dim3(nb), dim3(GEMM_THREADS), gemm_level_smem_bytes(nqt), ctx->stream, g);
    else hipLaunchKernelGGL(gemm_level_kernel<false>, dim3(nb), dim3(GEMM_THREADS), gemm_level_smem_bytes(nqt), ctx->stream, g);
}

}  // namespace smt
