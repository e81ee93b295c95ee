// ivfpq_build.hip -- building and extending the IVF index: coarse k-means on the MFMA pipe, quantiser training (global PQ or per-list
// PCA), encoding, inverted lists; smt_ivfpq_build / ivfpq_build_shared / smt_ivfpq_append.  Overview: ivfpq.h, DESIGN.md 4.6.
#include <rocprim/device/device_radix_sort.hpp>

#include "ivfpq.h"

namespace smt {

// ------------------------------------------------------------------ coarse assignment (MFMA)
struct AssignParams {
    const float *rows;        // corpus
    uint64_t n_points;        // points to assign
    uint64_t row_stride;      // point i = corpus row i * row_stride  (training sample: stride > 1)
    uint64_t n_rows_total;    // bound for reads
    const float *centroids;   // [nlist][256]  (BF16: the bf16 hi / lo split image of the centroids, same row size)
    const float *cnorm_half;  // [nlist] 0.5 * |c|^2
    uint32_t nlist;           // multiple of 32
    uint32_t *assign;         // [n_points]
};

// BF16: x . c from bf16 x 3 split products (mfma_tile.h; 16 x the f32 MFMA rate, error <= 1.5e-4 |x||c|): an assignment
// can only differ from the f32 one between two centroids that are equally good to that precision.
template <bool BF16>
__global__ void __launch_bounds__(GEMM_THREADS, 2) ivf_assign_kernel(AssignParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4 *s_c = reinterpret_cast<f32x4 *>(smem_raw);                 // [2][32][65] float4: centroid tiles
    float *s_cn = reinterpret_cast<float *>(s_c + 2 * QT_F4);        // [nlist]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const uint32_t nct = p.nlist / QT_ROWS;

    for (uint32_t c = threadIdx.x; c < p.nlist; c += GEMM_THREADS) s_cn[c] = p.cnorm_half[c];
    auto stage_load = [&](uint32_t ct, f32x4 (&r)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = threadIdx.x + u * GEMM_THREADS;
            r[u] = reinterpret_cast<const f32x4 *>(p.centroids + (size_t)(ct * QT_ROWS + (idx >> 6)) * 256)[idx & 63];
        }
    };
    auto stage_store = [&](int buf, const f32x4 (&r)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = threadIdx.x + u * GEMM_THREADS;
            s_c[buf * QT_F4 + (idx >> 6) * QT_STRIDE_F4 + (idx & 63)] = r[u];
        }
    };
    {
        f32x4 r[4];
        stage_load(0, r);
        stage_store(0, r);
    }
    __syncthreads();

    const uint64_t n_tiles = (p.n_points + 31) / 32;
    const uint64_t W = (uint64_t)gridDim.x * GEMM_WAVES;
    const uint64_t steps = (n_tiles + W - 1) / W;
    uint64_t it = (uint64_t)blockIdx.x * GEMM_WAVES + wave;
    int cur = 0;

    for (uint64_t step = 0; step < steps; ++step, it += W) {
        const bool has = it < n_tiles;
        const uint64_t p0 = (has ? it : 0) * 32;
        f32x4 A[BF16 ? 1 : 32];
        u32x4 Ah[BF16 ? 16 : 1], Al[BF16 ? 16 : 1];
        if (has) {
            const uint64_t pt = p0 + j;
            const bool ok = pt < p.n_points;
            if constexpr (BF16) {
                // lane (j, h): dims 16m + 8h .. + 7 of K-step m
                const f32x4 *src = reinterpret_cast<const f32x4 *>(p.rows + (ok ? pt * p.row_stride : 0) * 256) + 2 * h;
                f32x4 R[32];
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    R[2 * m] = ok ? src[4 * m] : (f32x4){0.f, 0.f, 0.f, 0.f};
                    R[2 * m + 1] = ok ? src[4 * m + 1] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int m = 0; m < 16; ++m) bf16_split8(R[2 * m], R[2 * m + 1], Ah[m], Al[m]);
            } else {
                const f32x4 *src = reinterpret_cast<const f32x4 *>(p.rows + (ok ? pt * p.row_stride : 0) * 256) + h;
#pragma unroll
                for (int m = 0; m < 32; ++m) A[m] = ok ? src[2 * m] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        unsigned long long best[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) best[r] = 0ull;

        for (uint32_t ct = 0; ct < nct; ++ct) {
            f32x4 nxt[4];
            const uint32_t ct_next = (ct + 1 == nct) ? 0 : ct + 1;
            stage_load(ct_next, nxt);
            if (has) {
                f32x16 acc;
                if constexpr (BF16) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                    const u32x4 *bq = reinterpret_cast<const u32x4 *>(s_c + cur * QT_F4 + j * QT_STRIDE_F4) + 2 * h;
#pragma unroll
                    for (int m = 0; m < 16; ++m) acc = mfma_bf16x3(Ah[m], Al[m], bq[4 * m], bq[4 * m + 1], acc);
                } else {
                    acc = mfma_tile_32x32x256(A, s_c + cur * QT_F4 + j * QT_STRIDE_F4 + h);
                }
                const uint32_t cid = ct * QT_ROWS + j;
                const float cn = s_cn[cid];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // arg-max of x.c - 0.5|c|^2 (== arg-min of |x - c|^2); ties -> smaller centroid id
                    const unsigned long long key =
                        ((unsigned long long)f32_orderable(acc[r] - cn) << 32) | (unsigned long long)(0xFFFFFFFFu - cid);
                    best[r] = key > best[r] ? key : best[r];
                }
            }
            stage_store(cur ^ 1, nxt);
            __syncthreads();
            cur ^= 1;
        }
        if (has) {
            // one reduction per row tile: max over the 32 lanes (columns) that share h
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                unsigned long long v = best[r];
#pragma unroll
                for (int off = 1; off < 32; off <<= 1) {
                    const unsigned long long o = __shfl_xor(v, off);
                    v = o > v ? o : v;
                }
                const uint64_t pt = p0 + acc_row(r, h);
                if (j == 0 && pt < p.n_points) p.assign[pt] = 0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull);
            }
        }
    }
}

__global__ void cnorm_half_kernel(const float *centroids, uint32_t nlist, float *out)
{
    const uint32_t c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= nlist) return;
    const int lane = threadIdx.x & 63;
    const f32x4 v = reinterpret_cast<const f32x4 *>(centroids + (size_t)c * 256)[lane];
    const float s = wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
    if (lane == 0) out[c] = 0.5f * s;
}

// one wave per point: sums[c][d] += x[d] in 2^-32 fixed point (integer atomics: order independent)
__global__ void ivf_accumulate_kernel(const float *rows, uint64_t n_points, uint64_t row_stride, const uint32_t *assign,
                                      long long *sums, unsigned int *counts)
{
    const uint64_t pt = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (pt >= n_points) return;
    const int lane = threadIdx.x & 63;
    const uint32_t c = assign[pt];
    const f32x4 v = reinterpret_cast<const f32x4 *>(rows + pt * row_stride * 256)[lane];
    long long *dst = sums + (size_t)c * 256 + lane * 4;
    atomicAdd(reinterpret_cast<unsigned long long *>(dst + 0), (unsigned long long)__double2ll_rn((double)v.x * FIXED_SCALE));
    atomicAdd(reinterpret_cast<unsigned long long *>(dst + 1), (unsigned long long)__double2ll_rn((double)v.y * FIXED_SCALE));
    atomicAdd(reinterpret_cast<unsigned long long *>(dst + 2), (unsigned long long)__double2ll_rn((double)v.z * FIXED_SCALE));
    atomicAdd(reinterpret_cast<unsigned long long *>(dst + 3), (unsigned long long)__double2ll_rn((double)v.w * FIXED_SCALE));
    if (lane == 0) atomicAdd(&counts[c], 1u);
}

__global__ void ivf_finalize_kernel(const long long *sums, const unsigned int *counts, uint32_t nlist, float *centroids)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlist * 256) return;
    const unsigned int n = counts[i >> 8];
    if (n) centroids[i] = (float)((double)sums[i] / FIXED_SCALE / (double)n);  // empty cluster keeps its centroid
}

// Shared-centroid builds (group.cpp): rank r seeds the lists l with l % n_ranks == r; every other list contributes
// nothing, so the all-reduce of (sums, counts) followed by ivf_finalize_kernel gives every rank the same start.
__global__ void ivf_seed_sums_kernel(const float *centroids, uint32_t nlist, uint32_t rank, uint32_t n_ranks, long long *sums,
                                     unsigned int *counts)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlist * 256) return;
    const uint32_t l = i >> 8;
    const bool mine = l % n_ranks == rank;
    sums[i] = mine ? __double2ll_rn((double)centroids[i] * FIXED_SCALE) : 0ll;
    if ((i & 255) == 0) counts[l] = mine ? 1u : 0u;
}

__global__ void gather_rows_kernel(const float *rows, uint64_t n, uint64_t row_stride, float *out)
{
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    reinterpret_cast<f32x4 *>(out + i * 256)[lane] = reinterpret_cast<const f32x4 *>(rows + i * row_stride * 256)[lane];
}

// ------------------------------------------------------------------ product quantiser
// Shared body: thread = (point, subspace) with 16 subspaces per pass; codebooks of the pass in LDS as
// [code][16][8] floats (lanes = subspaces read consecutive 32-B slots: conflict-free; lanes = points
// read the same address: broadcast).
struct PqParams {
    const float *rows;         // corpus
    const uint32_t *order;     // point i = corpus row order[i] (nullptr: row i * row_stride)
    uint64_t row_stride;
    const uint32_t *assign;    // coarse list of point i
    const float *centroids;    // [nlist][256]
    const float *codebooks;    // [32][256][8]
    uint64_t n_points;
    long long *sums;           // training: [32][256][8] fixed point (or nullptr)
    unsigned int *counts;      // training: [32][256]
    uint8_t *codes;            // encoding: [n_points][32] (or nullptr)
};

__global__ void __launch_bounds__(256) pq_assign_kernel(PqParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *s_cb = reinterpret_cast<float *>(smem_raw);  // [256][16][8] floats = 128 KiB
    const int sl = threadIdx.x & 15;   // subspace inside the pass
    const int pl = threadIdx.x >> 4;   // point inside the block (16 points per block)
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        for (int e = threadIdx.x; e < PQ_K * 16 * PQ_DSUB; e += blockDim.x) {
            const int d = e & 7, s = (e >> 3) & 15, code = e >> 7;
            s_cb[e] = p.codebooks[((size_t)(pass * 16 + s) * PQ_K + code) * PQ_DSUB + d];
        }
        __syncthreads();
        const int s = pass * 16 + sl;
        for (uint64_t base = (uint64_t)blockIdx.x * 16; base < p.n_points; base += (uint64_t)gridDim.x * 16) {
            const uint64_t i = base + pl;
            if (i >= p.n_points) continue;
            const uint64_t row = p.order ? (uint64_t)p.order[i] : i * p.row_stride;
            const f32x4 *x = reinterpret_cast<const f32x4 *>(p.rows + row * 256 + s * PQ_DSUB);
            const f32x4 *c = reinterpret_cast<const f32x4 *>(p.centroids + (size_t)p.assign[i] * 256 + s * PQ_DSUB);
            const f32x4 x0 = x[0], x1 = x[1], c0 = c[0], c1 = c[1];
            const float r[8] = {x0.x - c0.x, x0.y - c0.y, x0.z - c0.z, x0.w - c0.w,
                                x1.x - c1.x, x1.y - c1.y, x1.z - c1.z, x1.w - c1.w};
            float best = __builtin_inff();
            int best_code = 0;
#pragma unroll 4
            for (int code = 0; code < PQ_K; ++code) {
                const f32x4 *cb = reinterpret_cast<const f32x4 *>(s_cb + (code * 16 + sl) * PQ_DSUB);
                const f32x4 a = cb[0], b = cb[1];
                float d = 0.f, t;
                t = r[0] - a.x; d += t * t; t = r[1] - a.y; d += t * t; t = r[2] - a.z; d += t * t; t = r[3] - a.w; d += t * t;
                t = r[4] - b.x; d += t * t; t = r[5] - b.y; d += t * t; t = r[6] - b.z; d += t * t; t = r[7] - b.w; d += t * t;
                if (d < best) { best = d; best_code = code; }
            }
            if (p.codes) p.codes[i * PQ_M + s] = (uint8_t)best_code;
            if (p.sums) {
                long long *dst = p.sums + ((size_t)s * PQ_K + best_code) * PQ_DSUB;
#pragma unroll
                for (int d = 0; d < 8; ++d)
                    atomicAdd(reinterpret_cast<unsigned long long *>(dst + d), (unsigned long long)__double2ll_rn((double)r[d] * FIXED_SCALE));
                atomicAdd(&p.counts[s * PQ_K + best_code], 1u);
            }
        }
    }
}

__global__ void pq_finalize_kernel(const long long *sums, const unsigned int *counts, float *codebooks)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // over 32*256*8
    if (i >= PQ_M * PQ_K * PQ_DSUB) return;
    const unsigned int n = counts[i >> 3];
    if (n) codebooks[i] = (float)((double)sums[i] / FIXED_SCALE / (double)n);
}

// initial codebooks: residual sub-vectors of 256 evenly spaced training points
__global__ void pq_init_kernel(const float *rows, uint64_t row_stride, uint64_t n_points, const uint32_t *assign,
                               const float *centroids, float *codebooks)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // over 32*256*8
    if (i >= PQ_M * PQ_K * PQ_DSUB) return;
    const int d = i & 7, code = (i >> 3) & 255, s = i >> 11;
    const uint64_t pt = (uint64_t)code * (n_points / PQ_K);
    codebooks[i] = rows[pt * row_stride * 256 + s * PQ_DSUB + d] - centroids[(size_t)assign[pt] * 256 + s * PQ_DSUB + d];
}

// ------------------------------------------------------------------ inverted lists
__global__ void iota_kernel(uint32_t *v, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}

__global__ void list_offsets_kernel(const uint32_t *sorted_lists, uint64_t n, uint32_t nlist, uint64_t *offsets)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l > nlist) return;
    uint64_t lo = 0, hi = n;  // first index with sorted_lists[i] >= l
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (sorted_lists[mid] < l) lo = mid + 1; else hi = mid;
    }
    offsets[l] = lo;
}


// ------------------------------------------------------------------ per-list PCA codes (index kind 1)
// A global residual codebook (8 dims x 256 codes per sub-quantiser, above) spends its bits on all 256 coordinates
// of a residual alike: 25 % relative distortion per sub-vector, which on clustered data ranks the rows INSIDE a
// list poorly (10 M rows: recall@10 0.69 with 64 re-scored rows per list; 512 were needed for 0.98).  But the rows
// of one list differ from their centroid mostly inside a low-dimensional subspace that is the LIST's own.  Kind 1
// therefore gives every list its own orthonormal basis Q_l (the top 32 principal directions of its residuals:
// "locally optimised" product quantisation, Kalantidis & Avrithis 2014) and stores y = Q_l^T (x - c_l) with one
// 8-bit scalar quantiser per direction -- still m = 32 codes of 8 bits, 32 B per row, with dsub = 1 in the rotated
// space.  For unit rows  q.x = q.c_l + (Q_l^T q).y + (the part of q outside the subspace).(the part of x outside),
// so the ADC score is  base + sum_d w_d * code_d  with w = scale_l * Q_l^T q: 32 multiply-adds on the code bytes,
// no 32 KiB look-up table per query, no LDS gathers.
//
// lpca_train_kernel: one block per list, subspace (block power) iteration on the scatter matrix S = sum r r^T
// without ever forming it: Q <- orth(S Q) with S Q = sum over 32-row tiles of R^T (R Q^T)^T.  Deterministic: fixed
// tile order, fixed reduction trees, hash-seeded start vectors.
__device__ __forceinline__ float lp_hash_unit(uint32_t k, uint32_t d)
{
    uint32_t x = k * 0x9E3779B9u + d * 0x85EBCA6Bu + 0x165667B1u;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return (float)(int32_t)x * (1.0f / 2147483648.0f);
}

// sum over the block's 256 threads (4 waves), result in every thread; `red` = 4 floats of LDS
__device__ __forceinline__ float lp_block_sum(float v, float *red)
{
    const float w = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) lpca_train_kernel(const float *corpus, const uint32_t *ids, const uint64_t *offsets,
                                                          const float *centroids, uint32_t iters, float *basis, float *lscale)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *sQ = reinterpret_cast<float *>(smem_raw);            // [32][256]
    float *sR = sQ + LP_DIMS * 256;                             // [32][257]
    float *sZ = sR + LP_TILE * LP_RSTRIDE;                      // [32][33]
    float *sP = sZ + LP_TILE * 33;                              // [32] projections
    float *red = sP + LP_DIMS;                                  // [4]
    float *sLam = red + 4;                                      // [32] |S q_k| of the last iteration
    const uint32_t l = blockIdx.x;
    const int d = threadIdx.x;                                  // this thread's coordinate
    // The basis is fitted to an evenly spaced SAMPLE of the list (at most LP_TRAIN_ROWS rows): 32 directions of a residual
    // cloud that lives in a few dozen dimensions are pinned down by a few hundred rows, and the power iteration is the
    // dominant cost of the build (round 2: 250 of 362 ms at 10 M rows, 2.5 s at 100 M, every row of every list in every
    // iteration).  Every row is still ENCODED with the basis (lpca_encode_kernel).
    const uint64_t begin = offsets[l], n_list = offsets[l + 1] - begin;
    const uint64_t n = n_list < (uint64_t)LP_TRAIN_ROWS ? n_list : (uint64_t)LP_TRAIN_ROWS;   // rows the iteration sees
#pragma unroll
    for (int k = 0; k < LP_DIMS; ++k) sQ[k * 256 + d] = lp_hash_unit(k + 131u * l, d);
    if (d < LP_DIMS) sLam[d] = 0.0f;
    __syncthreads();

    // orthonormalise the 32 vectors held in sQ (thread d owns coordinate d of each): classical Gram-Schmidt with
    // re-orthogonalisation; a vector that vanishes (rank-deficient list) is replaced by a hash vector
    auto orthonormalise = [&](bool keep_norms) {
        for (int k = 0; k < LP_DIMS; ++k) {
            float v = sQ[k * 256 + d];
            float norm2_before = lp_block_sum(v * v, red);
            for (int pass = 0; pass < 3; ++pass) {
                for (int jj = 0; jj < k; ++jj) {
                    const float part = wave_sum(v * sQ[jj * 256 + d]);
                    if ((threadIdx.x & 63) == 0) sZ[jj * 4 + (threadIdx.x >> 6)] = part;   // sZ doubles as reduction scratch here
                }
                __syncthreads();
                float corr = 0.0f;
                for (int jj = 0; jj < k; ++jj) {
                    const float pj = (sZ[jj * 4] + sZ[jj * 4 + 1]) + (sZ[jj * 4 + 2] + sZ[jj * 4 + 3]);
                    corr += pj * sQ[jj * 256 + d];
                }
                __syncthreads();
                v -= corr;
                if (pass == 1) {
                    const float n2 = lp_block_sum(v * v, red);
                    if (n2 > 1e-30f && n2 > 1e-12f * norm2_before) break;   // a healthy direction: done after two passes
                    v = lp_hash_unit(977u + k + 131u * l, d);               // degenerate: restart from a hash vector (third pass cleans it)
                    norm2_before = 0.0f;
                }
            }
            const float n2 = lp_block_sum(v * v, red);
            if (keep_norms && d == 0) sLam[k] = norm2_before > 0.0f ? sqrtf(norm2_before) : 0.0f;  // |S q_k| ~ eigenvalue of the scatter matrix
            sQ[k * 256 + d] = n2 > 0.0f ? v * __frsqrt_rn(n2) : (d == k ? 1.0f : 0.0f);
            __syncthreads();
        }
    };
    orthonormalise(false);

    for (uint32_t it = 0; it < iters; ++it) {
        float acc[LP_DIMS];
#pragma unroll
        for (int k = 0; k < LP_DIMS; ++k) acc[k] = 0.0f;
        for (uint64_t t0 = 0; t0 < n; t0 += LP_TILE) {
            // residual tile: 32 rows x 256 dims (rows beyond the list are zero)
            for (int e = threadIdx.x; e < LP_TILE * 64; e += 256) {
                const int row = e >> 6, c4 = e & 63;
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (t0 + row < n) {
                    const uint64_t pick = n == n_list ? t0 + row : (t0 + row) * n_list / n;   // evenly spaced over the list
                    v = reinterpret_cast<const f32x4 *>(corpus + (uint64_t)ids[begin + pick] * 256)[c4];
                    const f32x4 c = reinterpret_cast<const f32x4 *>(centroids + (size_t)l * 256)[c4];
                    v -= c;
                }
                float *dst = sR + row * LP_RSTRIDE + 4 * c4;
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
            __syncthreads();
            {   // Z = R Q^T: thread -> (row, 4 directions)
                const int row = threadIdx.x >> 3, cg = threadIdx.x & 7;
                float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
                const float *r = sR + row * LP_RSTRIDE;
                const float *q0 = sQ + (4 * cg) * 256;
#pragma unroll 8
                for (int i = 0; i < 256; ++i) {
                    const float ri = r[i];
                    z0 += ri * q0[i]; z1 += ri * q0[256 + i]; z2 += ri * q0[512 + i]; z3 += ri * q0[768 + i];
                }
                float *z = sZ + row * 33 + 4 * cg;
                z[0] = z0; z[1] = z1; z[2] = z2; z[3] = z3;
            }
            __syncthreads();
            // (S Q)[k][d] += sum_row R[row][d] Z[row][k]
#pragma unroll 4
            for (int row = 0; row < LP_TILE; ++row) {
                const float r = sR[row * LP_RSTRIDE + d];
                const float *z = sZ + row * 33;
#pragma unroll
                for (int k = 0; k < LP_DIMS; ++k) acc[k] += r * z[k];
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < LP_DIMS; ++k) sQ[k * 256 + d] = acc[k];
        __syncthreads();
        orthonormalise(it + 1 == iters);
    }
#pragma unroll
    for (int k = 0; k < LP_DIMS; ++k) basis[((size_t)l * LP_DIMS + k) * 256 + d] = sQ[k * 256 + d];
    if (d < LP_DIMS) {
        // coefficient y_k has variance lambda_k / n: one 8-bit scalar quantiser per direction, range +-4 sigma
        const float lk = sLam[d];
        const float sigma = n > 0 ? sqrtf(lk / (float)n) : 0.0f;
        lscale[(size_t)l * LP_DIMS + d] = fmaxf(4.0f * sigma, 1e-12f) / 127.0f;
    }
}

// one wave per row (list order): code_k = clamp(round(Q_l[k] . (x - c_l) / scale_l[k])).  Grid-stride over the rows: a launch
// carries at most 2^32 - 1 work-items per dimension (the AQL packet's grid size is a u32 count of work-items), and 64 lanes
// per row pass that at 67 M rows -- the first version launched n * 64 threads and, at config c5's 100 M rows, silently
// encoded the first third of them only (recall 0.35; profiles/r03_ivf_sweep_100M_before_encode_fix.json is the run before, profiles/r03_ivfpq_100M.json the one after).
constexpr unsigned LPCA_ENCODE_MAX_BLOCKS = 1u << 20;   // x 4 waves: rows per pass of the grid
__global__ void __launch_bounds__(256) lpca_encode_kernel(const float *corpus, const uint32_t *ids, const uint32_t *sorted_lists,
                                                           uint64_t n, const float *centroids, const float *basis, const float *lscale,
                                                           uint8_t *codes)
{
    const int lane = threadIdx.x & 63;
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t pos = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); pos < n; pos += n_waves) {
        const uint32_t l = sorted_lists[pos];
        f32x4 r = reinterpret_cast<const f32x4 *>(corpus + (uint64_t)ids[pos] * 256)[lane];
        r -= reinterpret_cast<const f32x4 *>(centroids + (size_t)l * 256)[lane];
        const f32x4 *B = reinterpret_cast<const f32x4 *>(basis + (size_t)l * LP_DIMS * 256);
        uint32_t mine = 0;
#pragma unroll 4
        for (int k = 0; k < LP_DIMS; ++k) {
            const f32x4 b = B[k * 64 + lane];
            const float y = wave_sum(r.x * b.x + r.y * b.y + r.z * b.z + r.w * b.w);
            const float c = fminf(fmaxf(rintf(y / lscale[(size_t)l * LP_DIMS + k]), -127.0f), 127.0f);
            if (lane == k) mine = (uint32_t)(uint8_t)(int8_t)(int)c;
        }
        if (lane < LP_DIMS) codes[pos * PQ_M + lane] = (uint8_t)mine;
    }
}
}  // namespace smt

using namespace smt;

namespace {
using DevBuf = smt::IvfDevBuf;
inline int dev_alloc(DevBuf &b, size_t bytes) { return smt::ivf_dev_alloc(b, bytes); }
inline int compute_max_list(smt_ivfpq *ix) { return smt::ivf_compute_max_list(ix); }

constexpr size_t PQ_SMEM = (size_t)PQ_K * 16 * PQ_DSUB * 4;

constexpr size_t LPCA_SMEM = (size_t)(LP_DIMS * 256 + LP_TILE * LP_RSTRIDE + LP_TILE * 33 + LP_DIMS + 4 + LP_DIMS) * 4 + 64;

size_t assign_smem(uint32_t nlist) { return (size_t)2 * QT_F4 * 16 + (size_t)nlist * 4 + 64; }

int run_assign(smt_ctx *ctx, const float *rows, uint64_t n_points, uint64_t stride, uint64_t n_rows_total, const smt_ivfpq *ix,
               uint32_t *d_assign)
{
    if (!(ctx->attr_done & ATTR_IVF_ASSIGN)) {  // per context == per device
        IVF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ivf_assign_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        IVF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ivf_assign_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        IVF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pq_assign_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        ctx->attr_done |= ATTR_IVF_ASSIGN;
    }
    hipLaunchKernelGGL(cnorm_half_kernel, dim3((ix->nlist + 3) / 4), dim3(256), 0, ctx->stream, ix->d_centroids, ix->nlist,
                       ix->d_cnorm_half);
    AssignParams a;
    a.rows = rows;
    a.n_points = n_points;
    a.row_stride = stride;
    a.n_rows_total = n_rows_total;
    a.centroids = ix->d_centroids;
    a.cnorm_half = ix->d_cnorm_half;
    a.nlist = ix->nlist;
    a.assign = d_assign;
    const uint64_t tiles = (n_points + 31) / 32;
    const int blocks = (int)std::min<uint64_t>((uint64_t)ctx->num_cus, (tiles + GEMM_WAVES - 1) / GEMM_WAVES);
    if (ctx->tune.gemm_bf16x3) {
        // the centroids' split image lives in the context's scratch for the duration of this launch
        int rc = smt::ensure_scratch(ctx, (size_t)ix->nlist * 1024);
        if (rc) return rc;
        uint32_t *split = reinterpret_cast<uint32_t *>(ctx->d_scratch);
        if ((rc = smt::launch_split_rows_bf16(ctx, ix->d_centroids, ix->nlist, ix->nlist, split))) return rc;
        a.centroids = reinterpret_cast<const float *>(split);
        hipLaunchKernelGGL(ivf_assign_kernel<true>, dim3(blocks), dim3(GEMM_THREADS), assign_smem(ix->nlist), ctx->stream, a);
    } else {
        hipLaunchKernelGGL(ivf_assign_kernel<false>, dim3(blocks), dim3(GEMM_THREADS), assign_smem(ix->nlist), ctx->stream, a);
    }
    IVF_HIP(hipGetLastError());
    return SMT_OK;
}

double ms_since(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return (double)ms;
}

}  // namespace

extern "C" {

int smt_ivfpq_build(smt_corpus *corpus, const smt_ivfpq_params *prm, smt_ivfpq **out)
try {
    return smt::ivfpq_build_shared(corpus, prm, nullptr, out);
} catch (...) { return smt::api_catch(); }

}  // extern "C"

// The build proper.  `share` (or nullptr) makes it one rank of a data-parallel build over a row-sharded corpus
// (SURVEY 8e "C2"): after every accumulation of the coarse k-means -- on this rank's sample of ITS rows -- the
// fixed-point centroid sums and the counts are summed over the ranks (share->allreduce, enqueued on the context's
// stream: ncclAllReduce), so every rank finalises the SAME centroids and the lists mean the same thing on every
// shard; rows are then assigned, sorted and encoded locally (per-list PCA bases are fitted to the rank's own rows).
int smt::ivfpq_build_shared(smt_corpus *corpus, const smt_ivfpq_params *prm, const smt::IvfBuildShare *share, smt_ivfpq **out)
{
    SMT_REQUIRE(corpus && prm && out, "null argument");
    *out = nullptr;
    smt_ctx *ctx = corpus->ctx;
    IVF_HIP(hipSetDevice(ctx->device));
    { int rc_drain = smt::drain_async(ctx); if (rc_drain) return rc_drain; }
    const uint64_t N = corpus->rows;
    SMT_REQUIRE(prm->m == PQ_M && prm->nbits == 8, "this build supports m = 32 sub-quantisers of 8 bits");
    SMT_REQUIRE(prm->local_pca <= 1, "local_pca must be 0 or 1");
    SMT_REQUIRE(prm->reserved == 0, "smt_ivfpq_params.reserved must be 0 (it was `refine`, the int8 refinement stage removed in round 3)");
    const bool lpca = prm->local_pca == 1;
    SMT_REQUIRE(prm->nlist >= 32 && prm->nlist <= PROBE_MAX_LISTS && prm->nlist % 32 == 0, "nlist must be a multiple of 32 in [32, 4096]");
    SMT_REQUIRE(N >= (uint64_t)prm->nlist && N < 0xFFFFFFFFull, "corpus (shard) needs at least nlist rows");
    const uint32_t nlist = prm->nlist;
    const uint32_t iters = prm->train_iters ? prm->train_iters : 10;
    uint64_t S = prm->train_sample ? prm->train_sample : (uint64_t)64 * nlist;
    S = std::max<uint64_t>(std::min<uint64_t>(S, N), nlist);
    const uint64_t stride = N / S;  // evenly spaced sample

    smt_ivfpq *ix = new (std::nothrow) smt_ivfpq();
    if (!ix) { smt::set_error("out of host memory"); return SMT_E_NOMEM; }
    std::unique_ptr<smt_ivfpq, void (*)(smt_ivfpq *)> guard(ix, smt_ivfpq_destroy);
    ix->corpus = corpus;
    ix->device = corpus->ctx->device;
    ix->n_rows = N;
    ix->nlist = nlist;
    ix->kind = lpca ? 1u : 0u;
    // ---- set-up: every allocation of the coarse stage.  In a shared build the ranks then AGREE on its outcome, so that a
    // rank that ran out of memory takes the others with it instead of leaving them inside the first all-reduce.
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    DevBuf b_assign, b_sums, b_counts;
    int rc = [&]() -> int {
        // the index ranks rows as they are, not their directions: unit (or zero) rows only (domain.hip); part of the set-up so
        // that, in a shared build, a rank holding other rows takes the others with it
        if (int rcu = smt::require_unit_rows(ctx, corpus->d_rows, N, "smt_ivfpq_build")) return rcu;
        if (lpca) {
            IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_basis), (size_t)nlist * LP_DIMS * 256 * 4));
            IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_lscale), (size_t)nlist * LP_DIMS * 4));
        }
        IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_centroids), (size_t)nlist * 256 * 4));
        IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_cnorm_half), (size_t)nlist * 4));
        IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_codebooks), (size_t)PQ_M * PQ_K * PQ_DSUB * 4));
        IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_codes), (size_t)N * PQ_M));
        IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_ids), (size_t)N * 4));
        IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_offsets), (size_t)(nlist + 1) * 8));
        for (auto &e : ev) IVF_HIP(hipEventCreate(&e));
        int rc2;
        if ((rc2 = dev_alloc(b_assign, (size_t)std::max(S, N) * 4))) return rc2;
        if ((rc2 = dev_alloc(b_sums, (size_t)std::max<uint64_t>((uint64_t)nlist * 256, (uint64_t)PQ_M * PQ_K * PQ_DSUB) * 8))) return rc2;
        if ((rc2 = dev_alloc(b_counts, (size_t)std::max<uint64_t>(nlist, (uint64_t)PQ_M * PQ_K) * 4))) return rc2;
        return SMT_OK;
    }();
    if (share && share->agree) rc = share->agree(share->user, rc);
    if (rc) return rc;
    IVF_HIP(hipEventRecord(ev[0], ctx->stream));

    // ---- coarse k-means on the sample.  init: nlist evenly spaced sample points
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((nlist * 64 + 255) / 256)), dim3(256), 0, ctx->stream, corpus->d_rows,
                       (uint64_t)nlist, stride * (S / nlist), ix->d_centroids);
    if (share) {  // the same starting centroids on every rank: each rank seeds the lists it owns (l % n_ranks == rank)
        hipLaunchKernelGGL(ivf_seed_sums_kernel, dim3((nlist * 256 + 255) / 256), dim3(256), 0, ctx->stream, ix->d_centroids, nlist,
                           share->rank, share->n_ranks, b_sums.as<long long>(), b_counts.as<unsigned int>());
        if ((rc = share->allreduce(share->user, b_sums.as<long long>(), (size_t)nlist * 256, b_counts.as<unsigned int>(), nlist))) return rc;
        hipLaunchKernelGGL(ivf_finalize_kernel, dim3((nlist * 256 + 255) / 256), dim3(256), 0, ctx->stream, b_sums.as<long long>(),
                           b_counts.as<unsigned int>(), nlist, ix->d_centroids);
    }
    for (uint32_t it = 0; it < iters; ++it) {
        if ((rc = run_assign(ctx, corpus->d_rows, S, stride, N, ix, b_assign.as<uint32_t>()))) return rc;
        IVF_HIP(hipMemsetAsync(b_sums.p, 0, (size_t)nlist * 256 * 8, ctx->stream));
        IVF_HIP(hipMemsetAsync(b_counts.p, 0, (size_t)nlist * 4, ctx->stream));
        hipLaunchKernelGGL(ivf_accumulate_kernel, dim3((unsigned)((S * 64 + 255) / 256)), dim3(256), 0, ctx->stream, corpus->d_rows, S,
                           stride, b_assign.as<uint32_t>(), b_sums.as<long long>(), b_counts.as<unsigned int>());
        if (share && (rc = share->allreduce(share->user, b_sums.as<long long>(), (size_t)nlist * 256, b_counts.as<unsigned int>(), nlist)))
            return rc;
        hipLaunchKernelGGL(ivf_finalize_kernel, dim3((nlist * 256 + 255) / 256), dim3(256), 0, ctx->stream, b_sums.as<long long>(),
                           b_counts.as<unsigned int>(), nlist, ix->d_centroids);
    }
    IVF_HIP(hipEventRecord(ev[1], ctx->stream));

    // ---- product quantiser on the sample's residuals (kind 0; kind 1 trains per-list bases after the lists exist)
    if (!lpca) {
    if ((rc = run_assign(ctx, corpus->d_rows, S, stride, N, ix, b_assign.as<uint32_t>()))) return rc;
    hipLaunchKernelGGL(pq_init_kernel, dim3((PQ_M * PQ_K * PQ_DSUB + 255) / 256), dim3(256), 0, ctx->stream, corpus->d_rows, stride, S,
                       b_assign.as<uint32_t>(), ix->d_centroids, ix->d_codebooks);
    for (uint32_t it = 0; it < iters; ++it) {
        IVF_HIP(hipMemsetAsync(b_sums.p, 0, (size_t)PQ_M * PQ_K * PQ_DSUB * 8, ctx->stream));
        IVF_HIP(hipMemsetAsync(b_counts.p, 0, (size_t)PQ_M * PQ_K * 4, ctx->stream));
        PqParams q;
        q.rows = corpus->d_rows;
        q.order = nullptr;
        q.row_stride = stride;
        q.assign = b_assign.as<uint32_t>();
        q.centroids = ix->d_centroids;
        q.codebooks = ix->d_codebooks;
        q.n_points = S;
        q.sums = b_sums.as<long long>();
        q.counts = b_counts.as<unsigned int>();
        q.codes = nullptr;
        hipLaunchKernelGGL(pq_assign_kernel, dim3((unsigned)std::min<uint64_t>((S + 15) / 16, (uint64_t)ctx->num_cus)), dim3(256),
                           PQ_SMEM, ctx->stream, q);
        hipLaunchKernelGGL(pq_finalize_kernel, dim3((PQ_M * PQ_K * PQ_DSUB + 255) / 256), dim3(256), 0, ctx->stream,
                           b_sums.as<long long>(), b_counts.as<unsigned int>(), ix->d_codebooks);
    }
    }
    IVF_HIP(hipEventRecord(ev[2], ctx->stream));

    // ---- assign every row, sort rows by list, encode in list order
    if ((rc = run_assign(ctx, corpus->d_rows, N, 1, N, ix, b_assign.as<uint32_t>()))) return rc;
    IVF_HIP(hipEventRecord(ev[3], ctx->stream));
    DevBuf b_iota, b_sorted_lists, b_temp;
    if ((rc = dev_alloc(b_iota, (size_t)N * 4))) return rc;
    if ((rc = dev_alloc(b_sorted_lists, (size_t)N * 4))) return rc;
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, b_iota.as<uint32_t>(), N);
    size_t temp_bytes = 0;
    IVF_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, b_assign.as<uint32_t>(), b_sorted_lists.as<uint32_t>(), b_iota.as<uint32_t>(),
                                      ix->d_ids, N, 0, 32, ctx->stream));
    if ((rc = dev_alloc(b_temp, temp_bytes))) return rc;
    IVF_HIP(rocprim::radix_sort_pairs(b_temp.p, temp_bytes, b_assign.as<uint32_t>(), b_sorted_lists.as<uint32_t>(), b_iota.as<uint32_t>(),
                                      ix->d_ids, N, 0, 32, ctx->stream));
    hipLaunchKernelGGL(list_offsets_kernel, dim3((nlist + 1 + 255) / 256), dim3(256), 0, ctx->stream, b_sorted_lists.as<uint32_t>(), N, nlist,
                       ix->d_offsets);
    if (lpca) {
        // per-list PCA bases (subspace iteration over each list's residuals), then the 8-bit codes in list order
        if (!(ctx->attr_done & ATTR_IVF_LPCA)) {
            IVF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(lpca_train_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        160 * 1024));
            ctx->attr_done |= ATTR_IVF_LPCA;
        }
        const uint32_t pca_iters = std::max<uint32_t>(4u, std::min<uint32_t>(iters, 6u));
        hipLaunchKernelGGL(lpca_train_kernel, dim3(nlist), dim3(256), LPCA_SMEM, ctx->stream, corpus->d_rows, ix->d_ids, ix->d_offsets,
                           ix->d_centroids, pca_iters, ix->d_basis, ix->d_lscale);
        hipLaunchKernelGGL(lpca_encode_kernel, dim3((unsigned)std::min<uint64_t>((N + 3) / 4, LPCA_ENCODE_MAX_BLOCKS)), dim3(256), 0, ctx->stream, corpus->d_rows, ix->d_ids,
                           b_sorted_lists.as<uint32_t>(), N, ix->d_centroids, ix->d_basis, ix->d_lscale, ix->d_codes);
    } else {
        PqParams q;
        q.rows = corpus->d_rows;
        q.order = ix->d_ids;
        q.row_stride = 1;
        q.assign = b_sorted_lists.as<uint32_t>();
        q.centroids = ix->d_centroids;
        q.codebooks = ix->d_codebooks;
        q.n_points = N;
        q.sums = nullptr;
        q.counts = nullptr;
        q.codes = ix->d_codes;
        hipLaunchKernelGGL(pq_assign_kernel, dim3((unsigned)std::min<uint64_t>((N + 15) / 16, (uint64_t)ctx->num_cus)), dim3(256),
                           PQ_SMEM, ctx->stream, q);
    }
    hipLaunchKernelGGL(cnorm_half_kernel, dim3((nlist + 3) / 4), dim3(256), 0, ctx->stream, ix->d_centroids, nlist, ix->d_cnorm_half);
    IVF_HIP(hipEventRecord(ev[4], ctx->stream));
    IVF_HIP(hipGetLastError());
    IVF_HIP(hipStreamSynchronize(ctx->stream));
    ix->build_ms[0] = ms_since(ev[0], ev[1]);
    ix->build_ms[1] = ms_since(ev[2], ev[3]);
    ix->build_ms[2] = ms_since(ev[1], ev[2]);
    ix->build_ms[3] = ms_since(ev[3], ev[4]);
    for (auto &e : ev) (void)hipEventDestroy(e);
    if ((rc = compute_max_list(ix))) return rc;
    *out = guard.release();
    return SMT_OK;
}

namespace smt {
// smt_ivfpq_append: merge n_new already-encoded rows (sorted by list) into the inverted lists.
// grid = nlist blocks: block l copies its old segment to its new place and appends its new rows behind it.
__global__ void __launch_bounds__(256) ivf_merge_lists_kernel(const uint64_t *old_off, const uint64_t *new_off, const uint32_t *old_ids,
                                                               const uint8_t *old_codes, const uint32_t *new_ids, const uint8_t *new_codes,
                                                               uint32_t *out_ids, uint8_t *out_codes, uint64_t *out_off, uint32_t nlist)
{
    const uint32_t l = blockIdx.x;
    const uint64_t ob = old_off[l], on = old_off[l + 1] - ob;
    const uint64_t nb = new_off[l], nn = new_off[l + 1] - nb;
    const uint64_t dst = ob + nb;                 // rows of earlier lists: old ones + new ones
    for (uint64_t i = threadIdx.x; i < on; i += blockDim.x) out_ids[dst + i] = old_ids[ob + i];
    for (uint64_t i = threadIdx.x; i < nn; i += blockDim.x) out_ids[dst + on + i] = new_ids[nb + i];
    const uint4 *oc = reinterpret_cast<const uint4 *>(old_codes + ob * PQ_M);
    const uint4 *nc = reinterpret_cast<const uint4 *>(new_codes + nb * PQ_M);
    uint4 *dc = reinterpret_cast<uint4 *>(out_codes + dst * PQ_M);
    for (uint64_t i = threadIdx.x; i < on * 2; i += blockDim.x) dc[i] = oc[i];
    for (uint64_t i = threadIdx.x; i < nn * 2; i += blockDim.x) dc[on * 2 + i] = nc[i];
    if (threadIdx.x == 0) {
        out_off[l] = dst;
        if (l + 1 == nlist) out_off[nlist] = old_off[nlist] + new_off[nlist];
    }
}
__global__ void iota_from_kernel(uint32_t *v, uint64_t n, uint32_t first)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = first + (uint32_t)i;
}
}  // namespace smt

extern "C" {

int smt_ivfpq_append(smt_ivfpq *ix, uint64_t *n_added)
try {
    SMT_REQUIRE(ix != nullptr, "index");
    smt_corpus *corpus = ix->corpus;
    smt_ctx *ctx = corpus->ctx;
    if (n_added) *n_added = 0;
    SMT_REQUIRE(corpus->rows >= ix->n_rows, "the corpus shrank since the index was built: rebuild");
    const uint64_t n_old = ix->n_rows, n_new = corpus->rows - n_old, N = corpus->rows;
    if (n_new == 0) return SMT_OK;
    SMT_REQUIRE(N < 0xFFFFFFFFull, "a shard holds fewer than 2^32-1 rows");
    IVF_HIP(hipSetDevice(ctx->device));
    { int rc_drain = smt::drain_async(ctx); if (rc_drain) return rc_drain; }
    const uint32_t nlist = ix->nlist;
    DevBuf b_assign, b_sorted, b_iota, b_ids, b_temp, b_codes, b_noff;
    int rc;
    if ((rc = smt::require_unit_rows(ctx, corpus->d_rows + (size_t)n_old * 256, n_new, "smt_ivfpq_append", n_old))) return rc;
    if ((rc = dev_alloc(b_assign, n_new * 4)) || (rc = dev_alloc(b_sorted, n_new * 4)) || (rc = dev_alloc(b_iota, n_new * 4)) ||
        (rc = dev_alloc(b_ids, n_new * 4)) || (rc = dev_alloc(b_codes, n_new * PQ_M)) || (rc = dev_alloc(b_noff, (size_t)(nlist + 1) * 8)))
        return rc;
    // nearest centroid of every new row (the MFMA assignment kernel of the build), then the new rows in list order
    const float *new_rows = corpus->d_rows + (size_t)n_old * 256;
    if ((rc = run_assign(ctx, new_rows, n_new, 1, n_new, ix, b_assign.as<uint32_t>()))) return rc;
    hipLaunchKernelGGL(iota_from_kernel, dim3((unsigned)((n_new + 255) / 256)), dim3(256), 0, ctx->stream, b_iota.as<uint32_t>(), n_new,
                       (uint32_t)n_old);
    size_t temp_bytes = 0;
    IVF_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, b_assign.as<uint32_t>(), b_sorted.as<uint32_t>(), b_iota.as<uint32_t>(),
                                      b_ids.as<uint32_t>(), n_new, 0, 32, ctx->stream));
    if ((rc = dev_alloc(b_temp, temp_bytes))) return rc;
    IVF_HIP(rocprim::radix_sort_pairs(b_temp.p, temp_bytes, b_assign.as<uint32_t>(), b_sorted.as<uint32_t>(), b_iota.as<uint32_t>(),
                                      b_ids.as<uint32_t>(), n_new, 0, 32, ctx->stream));
    hipLaunchKernelGGL(list_offsets_kernel, dim3((nlist + 1 + 255) / 256), dim3(256), 0, ctx->stream, b_sorted.as<uint32_t>(), n_new, nlist,
                       b_noff.as<uint64_t>());
    // encode with the EXISTING quantisers (no retraining: that is what makes this incremental)
    if (ix->kind == 1) {
        hipLaunchKernelGGL(lpca_encode_kernel, dim3((unsigned)std::min<uint64_t>((n_new + 3) / 4, LPCA_ENCODE_MAX_BLOCKS)), dim3(256), 0, ctx->stream, corpus->d_rows,
                           b_ids.as<uint32_t>(), b_sorted.as<uint32_t>(), n_new, ix->d_centroids, ix->d_basis, ix->d_lscale,
                           b_codes.as<uint8_t>());
    } else {
        PqParams q;
        q.rows = corpus->d_rows;
        q.order = b_ids.as<uint32_t>();
        q.row_stride = 1;
        q.assign = b_sorted.as<uint32_t>();
        q.centroids = ix->d_centroids;
        q.codebooks = ix->d_codebooks;
        q.n_points = n_new;
        q.sums = nullptr;
        q.counts = nullptr;
        q.codes = b_codes.as<uint8_t>();
        hipLaunchKernelGGL(pq_assign_kernel, dim3((unsigned)std::min<uint64_t>((n_new + 15) / 16, (uint64_t)ctx->num_cus)), dim3(256),
                           PQ_SMEM, ctx->stream, q);
    }
    // merged lists: list l = its old rows, then its new rows (row order inside a list stays ascending)
    uint32_t *ids2 = nullptr;
    uint8_t *codes2 = nullptr;
    uint64_t *off2 = nullptr;
    IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ids2), (size_t)N * 4));
    if (hipMalloc(reinterpret_cast<void **>(&codes2), (size_t)N * PQ_M) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&off2), (size_t)(nlist + 1) * 8) != hipSuccess) {
        (void)hipFree(ids2); if (codes2) (void)hipFree(codes2);
        smt::set_error("out of device memory while extending the index");
        return SMT_E_NOMEM;
    }
    hipLaunchKernelGGL(ivf_merge_lists_kernel, dim3(nlist), dim3(256), 0, ctx->stream, ix->d_offsets, b_noff.as<uint64_t>(), ix->d_ids,
                       ix->d_codes, b_ids.as<uint32_t>(), b_codes.as<uint8_t>(), ids2, codes2, off2, nlist);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(ids2); (void)hipFree(codes2); (void)hipFree(off2);
        smt::set_error("extending the index: %s", hipGetErrorString(e));
        return SMT_E_HIP;
    }
    (void)hipFree(ix->d_ids); (void)hipFree(ix->d_codes); (void)hipFree(ix->d_offsets);
    ix->d_ids = ids2; ix->d_codes = codes2; ix->d_offsets = off2;
    ix->n_rows = N;
    if (n_added) *n_added = n_new;
    return compute_max_list(ix);
} catch (...) { return smt::api_catch(); }

}  // extern "C"
