// ivfpq_kernels.hip -- IVF-PQ index over the resident corpus (BASELINE config 5).
//
// NO reference counterpart: the reference's workspace store is an exact scan (its "IVF_PQ"/"HNSW"
// labels are cosmetic).  The contract is therefore recall against this library's own
// exact path, not parity with reference code.  Candidates found through the index are ALWAYS
// re-ranked with the exact f64 distance of the exact path (final_select_kernel), so every returned
// (row, distance) pair is a true pair -- only membership of the top-k is approximate.
//
// Build (all on the GPU):
//   coarse k-means (nlist centroids, D=256, L2): assignment = f32-MFMA C x centroids^T with the same
//     corpus-stationary tiling as K3 (row tile in registers, centroid tiles through LDS); each lane
//     keeps a running arg-max per row over the centroids it sees, one cross-lane reduce per row tile;
//     update = fixed-point (2^-32) integer atomics => order-independent, bit-reproducible centroids;
//   product quantiser: m=32 subspaces x 8 dims x 256 codes trained on residuals (x - centroid),
//     codebooks staged in LDS as [code][subspace][8] (conflict-free for lanes = subspaces);
//   inverted lists: rocPRIM radix sort of (list id, row), codes written in list order (32 B/row).
// Query:
//   probe: 0.5|c|^2 - q.c for all (query, centroid) pairs on the MFMA pipe, then one wave per query selects the
//     nprobe smallest of its 4096 scores held in registers (bisection on the orderable bit pattern);
//   LUT[s][code] = <q_s, codebook[s][code]> (inner product: rows are unit-norm, so the ADC score
//     q.c_list + sum_s LUT[s][code_s] approximates cos(q, x)); 32 KiB per query, LDS resident;
//   ADC scan: one block per (query, probed list) streams 32-B codes (nprobe/nlist * N * 32 B per query),
//     32 LDS lookups per row; each wave keeps its best candidates by threshold selection in registers,
//     (optionally prunes them with an int8 copy of the rows,) and re-scores them against the full-precision rows;
//   select: the K2 select stage merges the per-list candidate lists and rescoring is EXACT.
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <memory>

#include "common.h"
#include "device_utils.h"
#include "mfma_tile.h"

namespace smt {

constexpr int PQ_M = 32;      // subspaces
constexpr int PQ_DSUB = 8;    // dims per subspace (256 / 32)
constexpr int PQ_K = 256;     // codes per subspace (8 bits)
constexpr double FIXED_SCALE = 4294967296.0;  // 2^32

__device__ __forceinline__ uint32_t f32_orderable(float f)
{
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);  // ascending u32 == ascending float
}

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

// ------------------------------------------------------------------ query: probe
constexpr int PROBE_THREADS = 1024;
constexpr int PROBE_MAX_LISTS = 4096;

struct ProbeParams {
    const float *queries;     // [nq][256]
    const float *centroids;   // [nlist][256]
    const float *cnorm_half;
    uint32_t nlist;           // <= 4096
    uint32_t nprobe;
    uint32_t *probe_list;     // [nq][nprobe]
    float *probe_dot;         // [nq][nprobe]  q . c
};

// Coarse probe, two kernels (the first version was one block per query: 4096 wave-level dot products against
// centroids re-read from L2 by every query, then a 78-stage bitonic sort of all 4096 keys -- 0.45 ms per 1000
// queries, as much as the ADC scan itself):
//   ivf_score_kernel   S[q][c] = 0.5|c|^2 - q.c for all (query, centroid) pairs on the MFMA pipe; one block per
//                      centroid tile (32 centroids staged in LDS once), its waves sweep the query tiles;
//   ivf_probe_select_kernel   one WAVE per query holds its nlist scores in registers (64 per lane), finds the
//                      nprobe-th smallest by bisection on the orderable bit pattern, and emits the nprobe lists
//                      (order is irrelevant downstream; ties go to the smaller list id).
__global__ void __launch_bounds__(GEMM_THREADS, 2) ivf_score_kernel(const float *queries, uint32_t nq, const float *centroids,
                                                                    const float *cnorm_half, uint32_t nlist, float *scores)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4 *s_c = reinterpret_cast<f32x4 *>(smem_raw);  // [32][65] float4: this block's centroid tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const uint32_t ct = blockIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = threadIdx.x + u * GEMM_THREADS;
        s_c[(idx >> 6) * QT_STRIDE_F4 + (idx & 63)] =
            reinterpret_cast<const f32x4 *>(centroids + (size_t)(ct * QT_ROWS + (idx >> 6)) * 256)[idx & 63];
    }
    const uint32_t cid = ct * QT_ROWS + j;
    const float cn = cnorm_half[cid];
    __syncthreads();
    const uint32_t n_tiles = (nq + 31) / 32;
    for (uint32_t tile = wave; tile < n_tiles; tile += GEMM_WAVES) {
        const uint32_t qrow = tile * 32 + j;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(queries + (size_t)(qrow < nq ? qrow : 0) * 256) + h;
        f32x4 A[32];
#pragma unroll
        for (int m = 0; m < 32; ++m) A[m] = src[2 * m];
        const f32x16 acc = mfma_tile_32x32x256(A, s_c + j * QT_STRIDE_F4 + h);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t row = tile * 32 + acc_row(r, h);
            if (row < nq) scores[(size_t)row * nlist + cid] = cn - acc[r];
        }
    }
}

constexpr int SEL_SLOTS = PROBE_MAX_LISTS / 64;  // scores per lane

__global__ void __launch_bounds__(256) ivf_probe_select_kernel(ProbeParams p, const float *scores, uint32_t nq)
{
    const int lane = threadIdx.x & 63;
    const uint32_t qi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (qi >= nq) return;  // wave-uniform
    const float *sc = scores + (size_t)qi * p.nlist;
    uint32_t v[SEL_SLOTS];
#pragma unroll
    for (int u = 0; u < SEL_SLOTS; ++u) {
        const uint32_t c = (uint32_t)u * 64u + (uint32_t)lane;
        v[u] = c < p.nlist ? f32_orderable(sc[c]) : 0xFFFFFFFFu;
    }
    // smallest T with #(v <= T) >= nprobe
    uint32_t lo = 0u, hi = 0xFFFFFFFFu;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        uint32_t cnt = 0;
#pragma unroll
        for (int u = 0; u < SEL_SLOTS; ++u) cnt += v[u] <= mid ? 1u : 0u;
        cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_sum_u32(cnt));
        if (cnt >= p.nprobe) hi = mid; else lo = mid + 1u;
    }
    const uint32_t T = lo;
    uint32_t n_lt = 0;
#pragma unroll
    for (int u = 0; u < SEL_SLOTS; ++u) n_lt += v[u] < T ? 1u : 0u;
    n_lt = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_sum_u32(n_lt));
    uint32_t pos_lt = 0, pos_eq = n_lt;  // wave-uniform write cursors: "< T" first, then ties in list-id order
    uint32_t *out_l = p.probe_list + (size_t)qi * p.nprobe;
    float *out_d = p.probe_dot + (size_t)qi * p.nprobe;
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int u = 0; u < SEL_SLOTS; ++u) {
        const uint32_t c = (uint32_t)u * 64u + (uint32_t)lane;
        const bool lt = v[u] < T, eq = v[u] == T && c < p.nlist;
        const unsigned long long m_lt = __ballot(lt), m_eq = __ballot(eq);
        uint32_t slot = 0xFFFFFFFFu;
        if (lt) slot = pos_lt + (uint32_t)__popcll(m_lt & below);
        else if (eq) slot = pos_eq + (uint32_t)__popcll(m_eq & below);
        if (slot < p.nprobe) {
            out_l[slot] = c;
            out_d[slot] = p.cnorm_half[c] - sc[c];  // q . c
        }
        pos_lt += (uint32_t)__popcll(m_lt);
        pos_eq += (uint32_t)__popcll(m_eq);
    }
}

// LUT[q][s][code] = <q_s, codebook[s][code]>; grid (nq, 32), 256 threads
__global__ void ivf_lut_kernel(const float *queries, const float *codebooks, float *lut)
{
    const uint32_t qi = blockIdx.x, s = blockIdx.y, code = threadIdx.x;
    const float *q = queries + (size_t)qi * 256 + s * PQ_DSUB;
    const float *cb = codebooks + ((size_t)s * PQ_K + code) * PQ_DSUB;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < PQ_DSUB; ++d) acc += q[d] * cb[d];
    lut[((size_t)qi * PQ_M + s) * PQ_K + code] = acc;
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
constexpr int LP_DIMS = 32;            // directions kept per list == code bytes per row
constexpr int LP_TILE = 32;            // residual rows per tile
constexpr int LP_RSTRIDE = 257;        // LDS row stride of the residual tile (conflict-free column walks)
constexpr int LP_TRAIN_ROWS = 768;     // rows of a list the basis is fitted to (evenly spaced sample)

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
    const float cd = centroids[(size_t)l * 256 + d];
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
// encoded the first third of them only (recall 0.35; profiles/r03_ivf_sweep_100M_20k_topics.json has before and after).
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

// w[pair][k] = scale_l[k] * (Q_l[k] . q) for every (query, probed list) pair; one wave per pair
__global__ void __launch_bounds__(256) lpca_project_kernel(const float *queries, const uint32_t *probe_list, uint64_t n_pairs,
                                                            uint32_t nprobe, const float *basis, const float *lscale, float *w)
{
    const uint64_t pair = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (pair >= n_pairs) return;
    const int lane = threadIdx.x & 63;
    const uint32_t l = probe_list[pair];
    const f32x4 q = reinterpret_cast<const f32x4 *>(queries + (pair / nprobe) * 256)[lane];
    const f32x4 *B = reinterpret_cast<const f32x4 *>(basis + (size_t)l * LP_DIMS * 256);
    float mine = 0.0f;
#pragma unroll 4
    for (int k = 0; k < LP_DIMS; ++k) {
        const f32x4 b = B[k * 64 + lane];
        const float y = wave_sum(q.x * b.x + q.y * b.y + q.z * b.z + q.w * b.w);
        if (lane == k) mine = y * lscale[(size_t)l * LP_DIMS + k];
    }
    if (lane < LP_DIMS) w[pair * LP_DIMS + lane] = mine;
}

// ------------------------------------------------------------------ query: ADC scan
struct AdcParams {
    const float *queries;
    const float *lut;          // [nq][32][256]  (kind 0)
    const float *lw;           // [nq][nprobe][32] per-pair weights of the per-list PCA codes (kind 1), or nullptr
    const uint32_t *probe_list;
    const float *probe_dot;
    uint32_t nprobe;
    const uint64_t *list_offsets;
    const uint8_t *codes;      // [N][32] in list order
    const uint32_t *ids;       // [N] corpus row of each code
    const float *corpus;       // full-precision rows for the in-kernel re-score
    uint32_t n_seg;            // blocks per (query, probed list): a list is cut into segments of seg_len codes, each
    uint32_t seg_len;          //   with its own shortlist -- the re-scored fraction of a LONG list stays what it is for a short one
    uint32_t shortlist;        // ADC candidates kept per WAVE (<= 64); 4 or 8 waves per (query, list segment)
    uint32_t kp;               // re-scored candidates emitted per (query, list)  (<= 64)
    key_t64 *lists;            // [nq][nprobe][kp]
};

template <int ADC_THREADS>
__global__ void __launch_bounds__(ADC_THREADS) ivf_adc_kernel(AdcParams p)
{
    __shared__ __attribute__((aligned(16))) float s_lut[PQ_M * PQ_K];  // 32 KiB
    __shared__ key_t64 s_keys[(ADC_THREADS / 64) * 64];
    const uint32_t pi = blockIdx.x / p.n_seg, seg = blockIdx.x % p.n_seg, qi = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kp = (int)p.kp;
    const int ks = (int)p.shortlist;
    key_t64 *out = p.lists + ((size_t)qi * p.nprobe * p.n_seg + blockIdx.x) * kp;
    {
        // this block's segment of the probed list; most lists are shorter than n_seg segments: leave an empty list
        const uint32_t l0 = p.probe_list[(size_t)qi * p.nprobe + pi];
        const uint64_t b0 = p.list_offsets[l0], e0 = p.list_offsets[l0 + 1];
        if (b0 + (uint64_t)seg * p.seg_len >= e0) {  // block-uniform
            if ((int)threadIdx.x < kp) out[threadIdx.x] = KEY_PAD;
            return;
        }
    }

    if (p.lw == nullptr) {
        const f32x4 *lsrc = reinterpret_cast<const f32x4 *>(p.lut + (size_t)qi * PQ_M * PQ_K);
        for (int e = threadIdx.x; e < PQ_M * PQ_K / 4; e += ADC_THREADS) reinterpret_cast<f32x4 *>(s_lut)[e] = lsrc[e];
    }
    // kind 1: the 32 weights of this (query, list) pair, block-uniform (scalar loads)
    float lw[PQ_M];
    if (p.lw != nullptr) {
        const float *src = p.lw + ((size_t)qi * p.nprobe + pi) * PQ_M;
#pragma unroll
        for (int k = 0; k < PQ_M; ++k) lw[k] = src[k];
    } else {
#pragma unroll
        for (int k = 0; k < PQ_M; ++k) lw[k] = 0.0f;
    }
    const f32x4 qv = reinterpret_cast<const f32x4 *>(p.queries + (size_t)qi * 256)[lane];
    const float a2 = wave_sum(qv.x * qv.x + qv.y * qv.y + qv.z * qv.z + qv.w * qv.w);
    const bool qz = a2 == 0.0f;
    const float rq = qz ? 0.0f : __frsqrt_rn(a2);
    const uint32_t list = p.probe_list[(size_t)qi * p.nprobe + pi];
    const float base = p.probe_dot[(size_t)qi * p.nprobe + pi];
    const uint64_t begin = p.list_offsets[list] + (uint64_t)seg * p.seg_len;
    const uint64_t list_end = p.list_offsets[list + 1];
    const uint64_t end = seg + 1 == p.n_seg ? list_end : min(list_end, begin + (uint64_t)p.seg_len);  // the last segment takes the rest
    __syncthreads();

    // wave-uniform insert of (cd, cr) into a lane-distributed sorted list of `cap` entries
    auto insert = [&](float cd, uint32_t cr, float &ld, uint32_t &lr, float &thr_d, uint32_t &thr_r, int cap) {
        if (cd < thr_d || (cd == thr_d && cr < thr_r)) {
            const bool less = (ld < cd) || (ld == cd && lr < cr);
            const int pos = __popcll(__ballot(less));
            const float sd = dpp_f<DPP_WAVE_SHR1>(ld);
            const uint32_t sr = dpp_u<DPP_WAVE_SHR1>(lr);
            if (lane > pos) { ld = sd; lr = sr; }
            else if (lane == pos) { ld = cd; lr = cr; }
            thr_d = readlane_f(ld, cap - 1);
            thr_r = (uint32_t)__builtin_amdgcn_readlane((int)lr, cap - 1);
        }
    };

    // ---- stage 1: ADC scan of the list's codes -> the wave's `ks` best approximate candidates (an unordered SET:
    // lane i < n_short ends up holding one of them in (ld, lr)).  A wave takes 64 x ADC_R codes per pass, keeps
    // their ADC distances in registers next to the set carried over from the previous pass, finds the ks-th
    // smallest by bisection on the distance bits (ballot-free: per-lane counts + one DPP sum per step) and
    // compacts the winners through LDS.  (The first version inserted candidates one at a time into a sorted
    // lane-distributed list: ~160 serial inserts per wave at ks = 64 -- that, not the re-score reads, was what
    // bounded this kernel.)
    constexpr int ADC_R = 8;
    float ld = __builtin_inff();       // carried set: lane i < n_carry holds a real entry
    uint32_t lr = 0xFFFFFFFFu;
    key_t64 *s_short = s_keys + wave * 64;  // per-wave compaction scratch (s_keys is reused by the block merge later)
    // (64-code groups are dealt to the waves round-robin, so every wave sees codes from the whole list: lists are in
    // row order and neighbours cluster -- contiguous 512-code chunks per wave cost a point of recall)
    for (uint64_t base_i = begin; base_i < end; base_i += (uint64_t)(ADC_THREADS / 64) * 64 * ADC_R) {
        uint32_t kd[ADC_R + 1], kpos[ADC_R + 1];  // orderable distance bits (0xFFFFFFFF = empty) and list positions
#pragma unroll
        for (int r = 0; r < ADC_R; ++r) {
            const uint64_t i = base_i + ((uint64_t)r * (ADC_THREADS / 64) + wave) * 64 + lane;
            kd[r] = 0xFFFFFFFFu;
            kpos[r] = 0xFFFFFFFFu;
            if (i < end) {
                const uint4 c0 = reinterpret_cast<const uint4 *>(p.codes + i * PQ_M)[0];
                const uint4 c1 = reinterpret_cast<const uint4 *>(p.codes + i * PQ_M)[1];
                const uint32_t w[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                float acc = base;
                if (p.lw != nullptr) {   // block-uniform branch: signed bytes times the pair's weights
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        acc += lw[4 * u + 0] * (float)(int8_t)(w[u] & 0xFF);
                        acc += lw[4 * u + 1] * (float)(int8_t)((w[u] >> 8) & 0xFF);
                        acc += lw[4 * u + 2] * (float)(int8_t)((w[u] >> 16) & 0xFF);
                        acc += lw[4 * u + 3] * (float)(int8_t)(w[u] >> 24);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        acc += s_lut[(4 * u + 0) * PQ_K + (w[u] & 0xFF)];
                        acc += s_lut[(4 * u + 1) * PQ_K + ((w[u] >> 8) & 0xFF)];
                        acc += s_lut[(4 * u + 2) * PQ_K + ((w[u] >> 16) & 0xFF)];
                        acc += s_lut[(4 * u + 3) * PQ_K + (w[u] >> 24)];
                    }
                }
                const float d = fmaxf(1.0f - acc * rq, 0.0f);  // rows are unit-norm (model2vec output), zero rows score ~0
                if (acc == acc) {                                // a NaN score never becomes a candidate
                    kd[r] = min(__float_as_uint(d), 0xFFFFFFFEu);  // d >= 0: the bit pattern orders like the value
                    kpos[r] = (uint32_t)i;                         // position in list order (codes / ids / int8 rows share it)
                }
            }
        }
        kd[ADC_R] = lr != 0xFFFFFFFFu ? __float_as_uint(ld) : 0xFFFFFFFFu;  // (ld keeps only the top 16 bits: enough here)
        kpos[ADC_R] = lr;
        uint32_t total = 0;
#pragma unroll
        for (int r = 0; r <= ADC_R; ++r) total += kd[r] != 0xFFFFFFFFu ? 1u : 0u;
        total = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_sum_u32(total));
        // The ADC distance is itself an approximation (error ~1e-2): its top 16 bits (relative step 2^-8 of the value)
        // are all the selection needs, which halves the bisection; ties in that bucket go by scan order.
#pragma unroll
        for (int r = 0; r <= ADC_R; ++r) kd[r] = kd[r] == 0xFFFFFFFFu ? 0xFFFFFFFFu : (kd[r] >> 16);
        uint32_t T = 0xFFFFFFFEu, need_eq = 0xFFFFFFFFu;  // winners: kd < T, plus the first need_eq entries with kd == T
        if (total > (uint32_t)ks) {
            uint32_t lo = 0u, hi = 0xFFFFu;               // smallest T with #(kd <= T) >= ks
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                uint32_t cnt = 0;
#pragma unroll
                for (int r = 0; r <= ADC_R; ++r) cnt += kd[r] <= mid ? 1u : 0u;
                cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_sum_u32(cnt));
                if (cnt >= (uint32_t)ks) hi = mid; else lo = mid + 1u;
            }
            T = lo;
            uint32_t n_lt = 0;
#pragma unroll
            for (int r = 0; r <= ADC_R; ++r) n_lt += kd[r] < T ? 1u : 0u;
            n_lt = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_sum_u32(n_lt));
            need_eq = (uint32_t)ks - n_lt;
        }
        // compaction: winners take consecutive LDS slots, then lane i reads slot i
        uint32_t n_out = 0, n_eq_seen = 0;
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int r = 0; r <= ADC_R; ++r) {
            const bool valid = kd[r] != 0xFFFFFFFFu;
            const bool lt = valid && kd[r] < T;
            const bool eq = valid && kd[r] == T;
            const unsigned long long m_eq = __ballot(eq);
            const bool eq_win = eq && (n_eq_seen + (uint32_t)__popcll(m_eq & below)) < need_eq;
            const unsigned long long m_win = __ballot(lt || eq_win);
            if (lt || eq_win) s_short[n_out + (uint32_t)__popcll(m_win & below)] = ((key_t64)kd[r] << 32) | kpos[r];
            n_out += (uint32_t)__popcll(m_win);
            n_eq_seen += (uint32_t)__popcll(m_eq);
        }
        __builtin_amdgcn_wave_barrier();
        const key_t64 mine = (uint32_t)lane < n_out ? reinterpret_cast<volatile key_t64 *>(s_short)[lane] : KEY_PAD;
        __builtin_amdgcn_wave_barrier();
        ld = mine != KEY_PAD ? __uint_as_float((uint32_t)(mine >> 32) << 16) : __builtin_inff();
        lr = mine != KEY_PAD ? (uint32_t)(mine & 0xFFFFFFFFull) : 0xFFFFFFFFu;
    }

    // (An int8 refinement stage between the two -- a 260 B/row copy of the rows ranking the shortlist so that only a few
    // candidates need their 1 KiB row -- was built in round 1, measured at +6 % queries/s for a 7x larger index, kept opt-in
    // for two rounds and removed in round 3.)
    const int n_short = __popcll(__ballot(lr != 0xFFFFFFFFu));  // the set sits in lanes 0..n_short-1
    unsigned long long go = n_short >= 64 ? ~0ull : ((1ull << n_short) - 1ull);  // lanes whose candidate is re-scored

    // ---- stage 2: re-score the survivors with the full-precision rows (coalesced 1 KiB loads, f32),
    //      keep the kp best; the select stage then recomputes those exactly in f64
    const uint32_t my_row = (lane < n_short && ((go >> lane) & 1ull)) ? p.ids[lr] : 0xFFFFFFFFu;  // one gather, before the loop
    float ld2 = __builtin_inff();
    uint32_t lr2 = 0xFFFFFFFFu;
    float thr2_d = __builtin_inff();
    uint32_t thr2_r = 0xFFFFFFFFu;
    while (go) {
        f32x4 c[4];
        uint32_t rr[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ok[u] = go != 0ull;
            const int src = ok[u] ? __ffsll((long long)go) - 1 : 0;
            if (ok[u]) go &= go - 1;
            rr[u] = (uint32_t)__builtin_amdgcn_readlane((int)my_row, src);
            c[u] = reinterpret_cast<const f32x4 *>(p.corpus + (uint64_t)(ok[u] ? rr[u] : 0u) * 256)[lane];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float b2 = wave_sum(c[u].x * c[u].x + c[u].y * c[u].y + c[u].z * c[u].z + c[u].w * c[u].w);
            const float ab = wave_sum(c[u].x * qv.x + c[u].y * qv.y + c[u].z * qv.z + c[u].w * qv.w);
            if (ok[u]) insert(dist_f32(ab, b2, rq, qz), rr[u], ld2, lr2, thr2_d, thr2_r, kp);
        }
    }

    // block merge of the wave lists (rank by counting), as in K2
    s_keys[wave * 64 + lane] = (lane < kp && lr2 != 0xFFFFFFFFu) ? make_key(ld2, lr2) : KEY_PAD;
    if ((int)threadIdx.x < kp) out[threadIdx.x] = KEY_PAD;
    __syncthreads();
    const key_t64 mine = s_keys[wave * 64 + lane];
    if (mine != KEY_PAD) {
        int rank = 0;
        for (int w = 0; w < ADC_THREADS / 64; ++w)
            for (int i = 0; i < kp; ++i) rank += (s_keys[w * 64 + i] < mine) ? 1 : 0;
        if (rank < kp) out[rank] = mine;
    }
}

}  // namespace smt

// ====================================================================== host side
using namespace smt;

struct smt_ivfpq {
    smt_corpus *corpus = nullptr;
    int device = -1;                // the corpus' GPU (recorded so that destroy never has to look at the corpus)
    uint64_t n_rows = 0;
    uint32_t nlist = 0;
    float *d_centroids = nullptr;   // [nlist][256]
    float *d_cnorm_half = nullptr;  // [nlist]
    float *d_codebooks = nullptr;   // [32][256][8]
    uint8_t *d_codes = nullptr;     // [N][32] list order
    uint32_t *d_ids = nullptr;      // [N]
    uint64_t *d_offsets = nullptr;  // [nlist+1]
    uint32_t kind = 0;              // 0: global residual codebooks (dsub 8); 1: per-list PCA basis + 8-bit scalar codes
    float *d_basis = nullptr;       // kind 1: [nlist][32][256]
    float *d_lscale = nullptr;      // kind 1: [nlist][32]
    uint64_t max_list = 0;          // longest inverted list (segments per probed list at query time)
    double build_ms[4] = {0, 0, 0, 0};  // coarse train, assign all, pq train, encode+lists
};

namespace {

#define IVF_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            smt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SMT_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};

int dev_alloc(DevBuf &b, size_t bytes)
{
    hipError_t e = hipMalloc(&b.p, bytes ? bytes : 16);
    if (e != hipSuccess) { smt::set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return SMT_E_NOMEM; }
    return SMT_OK;
}

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

static int compute_max_list(smt_ivfpq *ix)
{
    std::vector<uint64_t> off(ix->nlist + 1);
    IVF_HIP(hipMemcpy(off.data(), ix->d_offsets, off.size() * 8, hipMemcpyDeviceToHost));
    ix->max_list = 0;
    for (uint32_t l = 0; l < ix->nlist; ++l) ix->max_list = std::max<uint64_t>(ix->max_list, off[l + 1] - off[l]);
    return SMT_OK;
}

extern "C" {

void smt_ivfpq_destroy(smt_ivfpq *ix)
{
    if (!ix) return;
    // The index is documented to be destroyed BEFORE its corpus, but a caller that gets the order wrong (a garbage
    // collector at interpreter exit, an error path) must not turn that into a use-after-free: nothing here touches the
    // corpus or its context -- the device ordinal was recorded at build / load time, and hipDeviceSynchronize covers
    // whatever stream the index's last kernels ran on.
    if (ix->device >= 0) { (void)hipSetDevice(ix->device); (void)hipDeviceSynchronize(); }
    for (void *p : {(void *)ix->d_centroids, (void *)ix->d_cnorm_half, (void *)ix->d_codebooks, (void *)ix->d_codes, (void *)ix->d_ids,
                    (void *)ix->d_offsets, (void *)ix->d_basis, (void *)ix->d_lscale})
        if (p) (void)hipFree(p);
    delete ix;
}

int smt_ivfpq_build(smt_corpus *corpus, const smt_ivfpq_params *prm, smt_ivfpq **out)
{
    return smt::ivfpq_build_shared(corpus, prm, nullptr, out);
}

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

extern "C" {

int smt_ivfpq_info(const smt_ivfpq *ix, uint64_t *n_rows, uint32_t *nlist, uint64_t *index_bytes, double *build_ms4)
{
    SMT_REQUIRE(ix != nullptr, "index");
    if (n_rows) *n_rows = ix->n_rows;
    if (nlist) *nlist = ix->nlist;
    if (index_bytes)
        *index_bytes = (uint64_t)ix->n_rows * (PQ_M + 4) + (uint64_t)ix->nlist * 256 * 4 + (uint64_t)PQ_M * PQ_K * PQ_DSUB * 4 +
                       (uint64_t)(ix->nlist + 1) * 8 +
                       (ix->kind == 1 ? (uint64_t)ix->nlist * LP_DIMS * 257 * 4 : 0);
    if (build_ms4) for (int i = 0; i < 4; ++i) build_ms4[i] = ix->build_ms[i];
    return SMT_OK;
}

int smt_ivfpq_list_sizes(const smt_ivfpq *ix, uint64_t *sizes_host)
{
    SMT_REQUIRE(ix && sizes_host, "null argument");
    std::vector<uint64_t> off(ix->nlist + 1);
    IVF_HIP(hipSetDevice(ix->corpus->ctx->device));
    IVF_HIP(hipMemcpy(off.data(), ix->d_offsets, off.size() * 8, hipMemcpyDeviceToHost));
    for (uint32_t l = 0; l < ix->nlist; ++l) sizes_host[l] = off[l + 1] - off[l];
    return SMT_OK;
}

// Shared by the host and the device entry points.  queries: host pointer (queries_on_device = false, staged into
// the scratch) or device pointer; the k best (row, exact distance) pairs and the counts are written to the DEVICE
// buffers d_or [nq][top_k], d_od [nq][top_k], d_oc [nq] (d_oc may be null).
static int ivfpq_search_core(smt_ivfpq *ix, const float *queries, bool queries_on_device, uint32_t nq, uint32_t top_k, uint32_t nprobe,
                             uint32_t rerank, uint64_t row_base, uint64_t *d_or_user, double *d_od_user, uint64_t *d_oc_user,
                             uint64_t **d_or_out, size_t *out_bytes_contig, uint64_t out_stride = 0)
{
    smt_ctx *ctx = ix->corpus->ctx;
    SMT_REQUIRE(ix->corpus->rows >= ix->n_rows, "the corpus shrank after the index was built: rebuild");
    SMT_REQUIRE(nprobe >= 1 && nprobe <= ix->nlist && nprobe <= 512, "nprobe must be in [1, min(nlist, 512)]");
    SMT_REQUIRE(top_k <= 56, "top_k must be <= 56 for the IVF-PQ path");
    if (rerank == 0) rerank = 512;
    SMT_REQUIRE(rerank >= 4 && rerank <= 512, "rerank (full-precision re-scored ADC candidates per probed list) must be in [4, 512]");
    const int adc_waves = rerank > 256 ? 8 : 4;   // waves per (query, list segment) block
    const uint32_t shortlist = (rerank + adc_waves - 1) / adc_waves;  // per wave
    const uint32_t kp = top_k + 8;                // re-scored candidates handed to the exact select stage
    // A list longer than ADC_SEGMENT codes is scanned by several blocks, each with its own shortlist of `rerank`
    // candidates (config 5's 100 M rows over 4096 lists: 24 k codes per list -- one shortlist of 512 would re-score
    // 2 % of them and recall@10 drops to 0.75); the select stage takes at most 512 lists per query.
    constexpr uint64_t ADC_SEGMENT = 8192;
    // sized by the TYPICAL list (1.5 x the mean), not the longest: a block that finds its segment empty still costs a
    // launch slot (+0.3 ms per 1000 queries when every list got a second, almost always empty, segment); the last
    // segment of an unusually long list simply takes the rest
    const uint64_t typical = (ix->n_rows / std::max<uint32_t>(ix->nlist, 1u)) * 3 / 2;
    uint32_t n_seg = (uint32_t)std::max<uint64_t>(1, (typical + ADC_SEGMENT - 1) / ADC_SEGMENT);
    n_seg = std::min<uint32_t>(n_seg, std::max<uint32_t>(1u, 512u / nprobe));
    const uint32_t seg_len = (uint32_t)ADC_SEGMENT;

    // every temporary lives in the context's scratch (no hipMalloc/hipFree per call), results come back through
    // the pinned staging buffer
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_q = 0, b_q = al((size_t)nq * 256 * 4);
    const size_t o_pl = o_q + b_q, b_pl = al((size_t)nq * nprobe * 4);
    const size_t o_pd = o_pl + b_pl, b_pd = b_pl;
    const size_t o_lut = o_pd + b_pd, b_lut = al((size_t)nq * PQ_M * PQ_K * 4);
    const size_t o_lists = o_lut + b_lut, b_lists = al((size_t)nq * nprobe * n_seg * kp * 8);
    const size_t o_or = o_lists + b_lists, b_or = (size_t)nq * top_k * 8;   // rows | dist | counts contiguous: one D2H
    const size_t o_od = o_or + b_or, b_od = b_or;
    const size_t o_oc = o_od + b_od, b_oc = al((size_t)nq * 8);
    const size_t o_sc = o_oc + b_oc, b_sc = al((size_t)nq * ix->nlist * 4);
    const size_t o_lw = o_sc + b_sc, b_lw = ix->kind == 1 ? al((size_t)nq * nprobe * PQ_M * 4) : 0;
    int rc = smt::ensure_scratch(ctx, o_lw + b_lw);
    if (rc) return rc;
    char *base = reinterpret_cast<char *>(ctx->d_scratch);
    const float *d_q = queries;
    if (!queries_on_device) {
        IVF_HIP(hipMemcpyAsync(base + o_q, queries, (size_t)nq * 256 * 4, hipMemcpyHostToDevice, ctx->stream));
        d_q = reinterpret_cast<const float *>(base + o_q);
    }

    if (!(ctx->attr_done & ATTR_IVF_SCORE)) {
        IVF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ivf_score_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->attr_done |= ATTR_IVF_SCORE;
    }
    ProbeParams pp;
    pp.queries = d_q;
    pp.centroids = ix->d_centroids;
    pp.cnorm_half = ix->d_cnorm_half;
    pp.nlist = ix->nlist;
    pp.nprobe = nprobe;
    pp.probe_list = reinterpret_cast<uint32_t *>(base + o_pl);
    pp.probe_dot = reinterpret_cast<float *>(base + o_pd);
    float *d_scores = reinterpret_cast<float *>(base + o_sc);
    prof_begin(ctx, "ivf_probe");
    hipLaunchKernelGGL(ivf_score_kernel, dim3(ix->nlist / QT_ROWS), dim3(GEMM_THREADS), (size_t)QT_F4 * 16 + 64, ctx->stream, d_q, nq,
                       ix->d_centroids, ix->d_cnorm_half, ix->nlist, d_scores);
    hipLaunchKernelGGL(ivf_probe_select_kernel, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, pp, d_scores, nq);
    if (ix->kind == 1) {
        const uint64_t n_pairs = (uint64_t)nq * nprobe;
        hipLaunchKernelGGL(lpca_project_kernel, dim3((unsigned)((n_pairs * 64 + 255) / 256)), dim3(256), 0, ctx->stream, d_q, pp.probe_list,
                           n_pairs, nprobe, ix->d_basis, ix->d_lscale, reinterpret_cast<float *>(base + o_lw));
    } else {
        hipLaunchKernelGGL(ivf_lut_kernel, dim3(nq, PQ_M), dim3(PQ_K), 0, ctx->stream, d_q, ix->d_codebooks, reinterpret_cast<float *>(base + o_lut));
    }
    prof_end(ctx, "ivf_probe");
    AdcParams ap;
    ap.queries = d_q;
    ap.lut = reinterpret_cast<float *>(base + o_lut);
    ap.lw = ix->kind == 1 ? reinterpret_cast<const float *>(base + o_lw) : nullptr;
    ap.probe_list = pp.probe_list;
    ap.probe_dot = pp.probe_dot;
    ap.nprobe = nprobe;
    ap.list_offsets = ix->d_offsets;
    ap.codes = ix->d_codes;
    ap.ids = ix->d_ids;
    ap.corpus = ix->corpus->d_rows;
    ap.n_seg = n_seg;
    ap.seg_len = seg_len ? seg_len : 512;
    ap.shortlist = shortlist;
    ap.kp = kp;
    ap.lists = reinterpret_cast<key_t64 *>(base + o_lists);
    prof_begin(ctx, "ivf_adc");
    if (adc_waves == 8) hipLaunchKernelGGL(ivf_adc_kernel<512>, dim3(nprobe * n_seg, nq), dim3(512), 0, ctx->stream, ap);
    else hipLaunchKernelGGL(ivf_adc_kernel<256>, dim3(nprobe * n_seg, nq), dim3(256), 0, ctx->stream, ap);
    prof_end(ctx, "ivf_adc");
    IVF_HIP(hipGetLastError());
    uint64_t *d_or = d_or_user ? d_or_user : reinterpret_cast<uint64_t *>(base + o_or);
    double *d_od = d_od_user ? d_od_user : reinterpret_cast<double *>(base + o_od);
    uint64_t *d_oc = d_or_user ? d_oc_user : reinterpret_cast<uint64_t *>(base + o_oc);
    SelectArgs sel;  // no exactness certificate: the index is approximate by contract (f32_err = 0)
    sel.corpus = ix->corpus->d_rows;
    sel.queries = d_q;
    sel.nq = nq;
    sel.lists = ap.lists;
    sel.n_lists = nprobe * n_seg;
    sel.kp = kp;
    sel.list_stride = (uint64_t)nprobe * n_seg * kp;
    sel.k_out = top_k;
    sel.row_base = row_base;
    sel.out_rows = d_or;
    sel.out_dist = d_od;
    sel.out_counts = d_oc;
    sel.out_stride = out_stride;
    rc = launch_select(ctx, sel);
    if (rc) return rc;
    if (d_or_out) *d_or_out = d_or;
    if (out_bytes_contig) *out_bytes_contig = b_or + b_od + (size_t)nq * 8;
    return SMT_OK;
}

int smt_ivfpq_search(smt_ivfpq *ix, const float *queries, uint32_t nq, uint32_t top_k, uint32_t nprobe, uint32_t rerank,
                     uint64_t row_base, uint64_t *out_rows, double *out_dist, uint64_t *out_counts, uint64_t out_cap)
{
    SMT_REQUIRE(ix != nullptr, "index");
    SMT_REQUIRE(nq == 0 || (queries && out_rows && out_dist && out_counts), "null argument");
    smt_ctx *ctx = ix->corpus->ctx;
    IVF_HIP(hipSetDevice(ctx->device));
    { int rc_drain = smt::drain_async(ctx); if (rc_drain) return rc_drain; }
    if (nq == 0) return SMT_OK;
    for (uint32_t q = 0; q < nq; ++q) out_counts[q] = 0;
    if (top_k == 0) return SMT_OK;
    uint64_t *d_or = nullptr;
    size_t out_bytes = 0;
    int rc = ivfpq_search_core(ix, queries, false, nq, top_k, nprobe, rerank, row_base, nullptr, nullptr, nullptr, &d_or, &out_bytes);
    if (rc) return rc;
    const size_t b_or = (size_t)nq * top_k * 8, b_od = b_or;
    if ((rc = smt::ensure_pinned(ctx, out_bytes))) return rc;
    IVF_HIP(hipMemcpyAsync(ctx->h_pinned, d_or, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    IVF_HIP(hipStreamSynchronize(ctx->stream));
    const uint64_t *h_rows = reinterpret_cast<const uint64_t *>(ctx->h_pinned);
    const double *h_dist = reinterpret_cast<const double *>(reinterpret_cast<const char *>(ctx->h_pinned) + b_or);
    const uint64_t *h_cnt = reinterpret_cast<const uint64_t *>(reinterpret_cast<const char *>(ctx->h_pinned) + b_or + b_od);
    bool truncated = false;
    for (uint32_t q = 0; q < nq; ++q) {
        out_counts[q] = h_cnt[q];
        const uint64_t w = std::min<uint64_t>(h_cnt[q], out_cap);
        if (h_cnt[q] > out_cap) truncated = true;
        for (uint64_t i = 0; i < w; ++i) {
            out_rows[(size_t)q * out_cap + i] = h_rows[(size_t)q * top_k + i];
            out_dist[(size_t)q * out_cap + i] = h_dist[(size_t)q * top_k + i];
        }
    }
    if (truncated) { smt::set_error("out_cap smaller than the number of hits"); return SMT_E_TRUNCATED; }
    return SMT_OK;
}

int smt_ivfpq_search_device(smt_ivfpq *ix, const float *queries_dev, uint32_t nq, uint32_t top_k, uint32_t nprobe, uint32_t rerank,
                            uint64_t row_base, uint64_t *out_rows_dev, double *out_dist_dev)
{
    SMT_REQUIRE(ix != nullptr, "index");
    SMT_REQUIRE(nq == 0 || (queries_dev && out_rows_dev && out_dist_dev), "null argument");
    SMT_REQUIRE(top_k >= 1, "top_k");
    smt_ctx *ctx = ix->corpus->ctx;
    IVF_HIP(hipSetDevice(ctx->device));
    { int rc_drain = smt::drain_async(ctx); if (rc_drain) return rc_drain; }
    if (nq == 0) return SMT_OK;
    return ivfpq_search_core(ix, queries_dev, true, nq, top_k, nprobe, rerank, row_base, out_rows_dev, out_dist_dev, nullptr, nullptr, nullptr);
}

}  // extern "C"

// one shard's answer in the packed exchange layout of group.cpp: [nq][2][top_k] words (global rows | f64 bits)
int smt::ivfpq_search_packed(smt_ivfpq *ix, const float *queries_dev, uint32_t nq, uint32_t top_k, uint32_t nprobe, uint32_t rerank,
                             uint64_t row_base, uint64_t *packed_dev)
{
    SMT_REQUIRE(ix && queries_dev && packed_dev, "null argument");
    smt_ctx *ctx = ix->corpus->ctx;
    IVF_HIP(hipSetDevice(ctx->device));
    { int rc_drain = smt::drain_async(ctx); if (rc_drain) return rc_drain; }
    return ivfpq_search_core(ix, queries_dev, true, nq, top_k, nprobe, rerank, row_base, packed_dev,
                             reinterpret_cast<double *>(packed_dev + top_k), nullptr, nullptr, nullptr, (uint64_t)2 * top_k);
}

// ---------------------------------------------------------------- persistence
// File = 64-byte little-endian header + the six device arrays in a fixed order.  The index refers to
// rows of a corpus by position, so it is only valid beside the corpus it was built on: load checks
// the row count (the workspace store keeps both files in one directory and rebuilds on mismatch).
namespace {
struct IvfFileHeader {
    char magic[8];      // "SMTIVFP1"
    uint32_t nlist, m, nbits, dim;
    uint64_t n_rows;
    uint8_t pad[32];
};
static_assert(sizeof(IvfFileHeader) == 64, "header layout");

bool write_dev(FILE *f, const void *d, size_t bytes, hipStream_t st, std::vector<char> &buf)
{
    const size_t chunk = (size_t)64 << 20;
    if (buf.size() < std::min(bytes, chunk)) buf.resize(std::min(bytes, chunk));
    for (size_t o = 0; o < bytes; o += chunk) {
        const size_t n = std::min(chunk, bytes - o);
        if (hipMemcpyAsync(buf.data(), (const char *)d + o, n, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
        if (hipStreamSynchronize(st) != hipSuccess) return false;
        if (fwrite(buf.data(), 1, n, f) != n) return false;
    }
    return true;
}

bool read_dev(FILE *f, void *d, size_t bytes, hipStream_t st, std::vector<char> &buf)
{
    const size_t chunk = (size_t)64 << 20;
    if (buf.size() < std::min(bytes, chunk)) buf.resize(std::min(bytes, chunk));
    for (size_t o = 0; o < bytes; o += chunk) {
        const size_t n = std::min(chunk, bytes - o);
        if (fread(buf.data(), 1, n, f) != n) return false;
        if (hipMemcpyAsync((char *)d + o, buf.data(), n, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        if (hipStreamSynchronize(st) != hipSuccess) return false;
    }
    return true;
}
}  // namespace

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

namespace smt {
__global__ void count_ids_out_of_range_kernel(const uint32_t *ids, uint64_t n, uint32_t n_rows, unsigned int *bad)
{
    unsigned int mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        mine += ids[i] >= n_rows ? 1u : 0u;
    if (mine) atomicAdd(bad, mine);
}
}  // namespace smt

extern "C" {

int smt_ivfpq_append(smt_ivfpq *ix, uint64_t *n_added)
{
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
}

int smt_ivfpq_save(smt_ivfpq *ix, const char *path)
{
    SMT_REQUIRE(ix && path, "null argument");
    smt_ctx *ctx = ix->corpus->ctx;
    IVF_HIP(hipSetDevice(ctx->device));
    { int rc_drain = smt::drain_async(ctx); if (rc_drain) return rc_drain; }
    IVF_HIP(hipStreamSynchronize(ctx->stream));
    FILE *f = fopen(path, "wb");
    if (!f) { smt::set_error("cannot open '%s' for writing: %s", path, strerror(errno)); return SMT_E_IO; }
    IvfFileHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "SMTIVFP1", 8);
    h.nlist = ix->nlist;
    h.m = PQ_M;
    h.nbits = 8;
    h.dim = 256;
    h.n_rows = ix->n_rows;
    h.pad[0] = 0;  // (1 marked an index built with the int8 refinement copy, removed in round 3: load refuses such files)
    h.pad[1] = (uint8_t)ix->kind; // 1 = per-list PCA bases + scalar codes follow the codes
    std::vector<char> buf;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    ok = ok && write_dev(f, ix->d_centroids, (size_t)ix->nlist * 256 * 4, ctx->stream, buf);
    ok = ok && write_dev(f, ix->d_cnorm_half, (size_t)ix->nlist * 4, ctx->stream, buf);
    ok = ok && write_dev(f, ix->d_codebooks, (size_t)PQ_M * PQ_K * PQ_DSUB * 4, ctx->stream, buf);
    ok = ok && write_dev(f, ix->d_offsets, (size_t)(ix->nlist + 1) * 8, ctx->stream, buf);
    ok = ok && write_dev(f, ix->d_ids, (size_t)ix->n_rows * 4, ctx->stream, buf);
    ok = ok && write_dev(f, ix->d_codes, (size_t)ix->n_rows * PQ_M, ctx->stream, buf);
    if (ix->kind == 1) {
        ok = ok && write_dev(f, ix->d_basis, (size_t)ix->nlist * LP_DIMS * 256 * 4, ctx->stream, buf);
        ok = ok && write_dev(f, ix->d_lscale, (size_t)ix->nlist * LP_DIMS * 4, ctx->stream, buf);
    }
    if (fclose(f) != 0) ok = false;
    if (!ok) { smt::set_error("short write to '%s'", path); return SMT_E_IO; }
    return SMT_OK;
}

int smt_ivfpq_load(smt_corpus *corpus, const char *path, smt_ivfpq **out)
{
    SMT_REQUIRE(corpus && path && out, "null argument");
    *out = nullptr;
    smt_ctx *ctx = corpus->ctx;
    IVF_HIP(hipSetDevice(ctx->device));
    { int rc_drain = smt::drain_async(ctx); if (rc_drain) return rc_drain; }
    FILE *f = fopen(path, "rb");
    if (!f) { smt::set_error("cannot open '%s': %s", path, strerror(errno)); return SMT_E_IO; }
    std::unique_ptr<FILE, int (*)(FILE *)> fguard(f, fclose);
    IvfFileHeader h;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "SMTIVFP1", 8) != 0) {
        smt::set_error("'%s' is not an IVF-PQ index file", path);
        return SMT_E_IO;
    }
    if (h.m != PQ_M || h.nbits != 8 || h.dim != 256 || h.nlist < 32 || h.nlist > PROBE_MAX_LISTS || h.nlist % 32) {
        smt::set_error("'%s': unsupported index geometry (nlist %u, m %u, nbits %u, dim %u)", path, h.nlist, h.m, h.nbits, h.dim);
        return SMT_E_UNSUPPORTED;
    }
    if (h.n_rows > corpus->rows) {  // (fewer is fine: the index covers a prefix, smt_ivfpq_append takes in the rest)
        smt::set_error("'%s' indexes %llu rows but the corpus holds %llu: rebuild", path, (unsigned long long)h.n_rows,
                       (unsigned long long)corpus->rows);
        return SMT_E_INVALID;
    }
    smt_ivfpq *ix = new (std::nothrow) smt_ivfpq();
    if (!ix) { smt::set_error("out of host memory"); return SMT_E_NOMEM; }
    std::unique_ptr<smt_ivfpq, void (*)(smt_ivfpq *)> guard(ix, smt_ivfpq_destroy);
    ix->corpus = corpus;
    ix->device = corpus->ctx->device;
    ix->n_rows = h.n_rows;
    ix->nlist = h.nlist;
    const size_t N = (size_t)h.n_rows;
    IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_centroids), (size_t)h.nlist * 256 * 4));
    IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_cnorm_half), (size_t)h.nlist * 4));
    IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_codebooks), (size_t)PQ_M * PQ_K * PQ_DSUB * 4));
    IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_offsets), (size_t)(h.nlist + 1) * 8));
    IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_ids), std::max<size_t>(N * 4, 16)));
    IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_codes), std::max<size_t>(N * PQ_M, 16)));
    std::vector<char> buf;
    bool ok = read_dev(f, ix->d_centroids, (size_t)h.nlist * 256 * 4, ctx->stream, buf);
    ok = ok && read_dev(f, ix->d_cnorm_half, (size_t)h.nlist * 4, ctx->stream, buf);
    ok = ok && read_dev(f, ix->d_codebooks, (size_t)PQ_M * PQ_K * PQ_DSUB * 4, ctx->stream, buf);
    ok = ok && read_dev(f, ix->d_offsets, (size_t)(h.nlist + 1) * 8, ctx->stream, buf);
    ok = ok && read_dev(f, ix->d_ids, N * 4, ctx->stream, buf);
    ok = ok && read_dev(f, ix->d_codes, N * PQ_M, ctx->stream, buf);
    ix->kind = h.pad[1];
    if (ix->kind > 1) { smt::set_error("'%s': unknown index kind %u", path, ix->kind); return SMT_E_IO; }
    if (ok && ix->kind == 1) {
        IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_basis), (size_t)h.nlist * LP_DIMS * 256 * 4));
        IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_lscale), (size_t)h.nlist * LP_DIMS * 4));
        ok = ok && read_dev(f, ix->d_basis, (size_t)h.nlist * LP_DIMS * 256 * 4, ctx->stream, buf);
        ok = ok && read_dev(f, ix->d_lscale, (size_t)h.nlist * LP_DIMS * 4, ctx->stream, buf);
    }
    if (!ok) { smt::set_error("'%s' is truncated or unreadable", path); return SMT_E_IO; }
    if (h.pad[0] == 1) { smt::set_error("'%s' was built with the int8 refinement stage (removed): rebuild the index", path); return SMT_E_INVALID; }
    // the list table must be consistent with the row count, or the ADC kernel would read out of bounds
    std::vector<uint64_t> offs((size_t)h.nlist + 1);
    IVF_HIP(hipMemcpy(offs.data(), ix->d_offsets, offs.size() * 8, hipMemcpyDeviceToHost));
    bool sane = offs[0] == 0 && offs[h.nlist] == h.n_rows;
    for (uint32_t l = 0; sane && l < h.nlist; ++l) sane = offs[l] <= offs[l + 1];
    if (!sane) { smt::set_error("'%s': corrupt list offsets", path); return SMT_E_IO; }
    // ... and every stored row id must name a row of THIS corpus: the re-score gathers corpus rows by id (a stale or corrupt file would otherwise read out of bounds)
    if (N > 0) {
        int rc_s = smt::ensure_scratch(ctx, 64);
        if (rc_s) return rc_s;
        unsigned int *d_bad = reinterpret_cast<unsigned int *>(ctx->d_scratch);
        IVF_HIP(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
        hipLaunchKernelGGL(count_ids_out_of_range_kernel, dim3((unsigned)std::min<uint64_t>((N + 255) / 256, 65535)), dim3(256), 0,
                           ctx->stream, ix->d_ids, N, (uint32_t)std::min<uint64_t>(h.n_rows, 0xFFFFFFFFull), d_bad);
        unsigned int bad = 0;
        IVF_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
        IVF_HIP(hipStreamSynchronize(ctx->stream));
        if (bad) { smt::set_error("'%s': %u row ids outside the corpus (stale or corrupt index file)", path, bad); return SMT_E_IO; }
    }
    ix->max_list = 0;
    for (uint32_t l = 0; l < h.nlist; ++l) ix->max_list = std::max<uint64_t>(ix->max_list, offs[l + 1] - offs[l]);
    *out = guard.release();
    return SMT_OK;
}

}  // extern "C"
