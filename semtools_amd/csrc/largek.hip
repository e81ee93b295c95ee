// largek.hip -- top_k > 64: the rare "give me hundreds of lines" case.
// The reference sorts ALL N results and takes k (src/search/mod.rs:107-119); here every row's
// f32 key (distance bits << 32 | row) is written once (8 B/row on top of the 1 KiB/row read),
// rocPRIM radix-sorts the keys, the best k + guard rows are rescored exactly on the GPU and the
// host orders those few by (f64 distance, row).  Not a hot path: one extra pass over 8 B/row.
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"
#include "device_utils.h"

namespace smt {

struct AllKeysParams {
    const float *corpus;
    const float *query;
    uint64_t n_virtual;
    const smt_range *ranges;
    const uint64_t *prefix;
    uint32_t n_ranges;
    key_t64 *keys;  // [n_virtual]
};

__device__ __forceinline__ uint32_t map_virtual_lk(uint64_t v, const smt_range *ranges, const uint64_t *prefix,
                                                   uint32_t n_ranges)
{
    uint32_t lo = 0, hi = n_ranges;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (prefix[mid] <= v) lo = mid; else hi = mid;
    }
    return (uint32_t)(ranges[lo].begin + (v - prefix[lo]));
}

template <int U, bool FILTERED>
__global__ void __launch_bounds__(1024) scan_allkeys_kernel(AllKeysParams p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_block = blockDim.x >> 6;
    const uint64_t wave_global = (uint64_t)blockIdx.x * waves_per_block + wave;
    const uint64_t stride = (uint64_t)gridDim.x * waves_per_block * U;
    const f32x4 q = reinterpret_cast<const f32x4 *>(p.query)[lane];
    const float a2 = wave_sum(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const bool qz = (a2 == 0.0f);
    const float rq = qz ? 0.0f : __frsqrt_rn(a2);
    for (uint64_t v0 = wave_global * U; v0 < p.n_virtual; v0 += stride) {
        f32x4 c[U];
        uint32_t row[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            uint64_t v = v0 + j;
            if (v >= p.n_virtual) v = p.n_virtual - 1;
            row[j] = FILTERED ? map_virtual_lk(v, p.ranges, p.prefix, p.n_ranges) : (uint32_t)v;
            c[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p.corpus + (uint64_t)row[j] * 256) + lane);
        }
        key_t64 mine = KEY_PAD;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const float b2 = wave_sum(c[j].x * c[j].x + c[j].y * c[j].y + c[j].z * c[j].z + c[j].w * c[j].w);
            const float ab = wave_sum(c[j].x * q.x + c[j].y * q.y + c[j].z * q.z + c[j].w * q.w);
            float d = dist_f32(ab, b2, rq, qz);
            if (!(d == d)) d = __builtin_inff();  // NaN rows can never be selected: park them at the end
            if (lane == j) mine = make_key(d, row[j]);
        }
        if (lane < U && v0 + lane < p.n_virtual) p.keys[v0 + lane] = mine;
    }
}

__global__ void keys_to_rows_kernel(const key_t64 *keys, uint32_t *rows, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rows[i] = (uint32_t)(keys[i] & 0xFFFFFFFFull);
}

// Host-visible result: candidate rows (local) + exact distances, best-first by f32 key; the caller
// applies thresholds and the final (f64, row) order.
int launch_largek_candidates(smt_ctx *ctx, const float *corpus, const float *query_dev, const smt_range *ranges_dev,
                             const uint64_t *prefix_dev, uint32_t n_ranges, uint64_t n_virtual, uint64_t n_cand,
                             std::vector<uint32_t> &rows_out, std::vector<double> &dist_out, float *next_d32)
{
    SMT_REQUIRE(n_cand <= n_virtual, "candidate count");
    key_t64 *keys = nullptr, *sorted = nullptr;
    void *temp = nullptr;
    uint32_t *d_rows = nullptr;
    double *d_dist = nullptr;
    auto cleanup = [&]() {
        if (keys) (void)hipFree(keys);
        if (sorted) (void)hipFree(sorted);
        if (temp) (void)hipFree(temp);
        if (d_rows) (void)hipFree(d_rows);
        if (d_dist) (void)hipFree(d_dist);
    };
#define LK_CHECK(expr)                                                                         \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            set_error("%s failed: %s", #expr, hipGetErrorString(_e));                          \
            cleanup();                                                                         \
            return SMT_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)
    LK_CHECK(hipMalloc(reinterpret_cast<void **>(&keys), n_virtual * sizeof(key_t64)));
    LK_CHECK(hipMalloc(reinterpret_cast<void **>(&sorted), n_virtual * sizeof(key_t64)));
    AllKeysParams p;
    p.corpus = corpus;
    p.query = query_dev;
    p.n_virtual = n_virtual;
    p.ranges = ranges_dev;
    p.prefix = prefix_dev;
    p.n_ranges = n_ranges;
    p.keys = keys;
    const int blocks = 2 * ctx->num_cus, threads = 512;
    prof_begin(ctx, "scan");
    if (n_ranges) hipLaunchKernelGGL((scan_allkeys_kernel<8, true>), dim3(blocks), dim3(threads), 0, ctx->stream, p);
    else hipLaunchKernelGGL((scan_allkeys_kernel<8, false>), dim3(blocks), dim3(threads), 0, ctx->stream, p);
    prof_end(ctx, "scan");
    LK_CHECK(hipGetLastError());
    size_t temp_bytes = 0;
    LK_CHECK(rocprim::radix_sort_keys(nullptr, temp_bytes, keys, sorted, n_virtual, 0, 64, ctx->stream));
    LK_CHECK(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    LK_CHECK(rocprim::radix_sort_keys(temp, temp_bytes, keys, sorted, n_virtual, 0, 64, ctx->stream));
    LK_CHECK(hipMalloc(reinterpret_cast<void **>(&d_rows), n_cand * sizeof(uint32_t)));
    LK_CHECK(hipMalloc(reinterpret_cast<void **>(&d_dist), n_cand * sizeof(double)));
    hipLaunchKernelGGL(keys_to_rows_kernel, dim3((unsigned)((n_cand + 255) / 256)), dim3(256), 0, ctx->stream, sorted,
                       d_rows, n_cand);
    int rc = launch_rescore_rows(ctx, corpus, query_dev, d_rows, n_cand, d_dist);
    if (rc) { cleanup(); return rc; }
    rows_out.resize(n_cand);
    dist_out.resize(n_cand);
    LK_CHECK(hipMemcpyAsync(rows_out.data(), d_rows, n_cand * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    LK_CHECK(hipMemcpyAsync(dist_out.data(), d_dist, n_cand * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    key_t64 next_key = KEY_PAD;
    if (next_d32 && n_cand < n_virtual)
        LK_CHECK(hipMemcpyAsync(&next_key, sorted + n_cand, sizeof(key_t64), hipMemcpyDeviceToHost, ctx->stream));
    LK_CHECK(hipStreamSynchronize(ctx->stream));
    if (next_d32) {
        const uint32_t bits = (uint32_t)(next_key >> 32);
        float f = __builtin_inff();
        if (next_key != KEY_PAD) memcpy(&f, &bits, sizeof(f));
        *next_d32 = f;
    }
#undef LK_CHECK
    cleanup();
    return SMT_OK;
}

}  // namespace smt
