// threshold.hip -- K4: "every row with distance < max_distance" (src/search/mod.rs:88-89, 115-116: with
// a threshold the reference returns ALL hits, sorted by distance, row order on ties; top_k is ignored).
//
//   1. scan_threshold_kernel: the K2 streaming loop (one coalesced 1 KiB row per wave instruction, next
//      chunk prefetched, chunks claimed from the block's LDS counter).  A row passes the f32 prefilter
//      d32 < max_distance + 8e-6 (guard band: the f32 value may sit a few ulp off the f64 one).  Hits are
//      collected in a per-wave LDS buffer and appended with ONE global atomic per 256 hits -- the first
//      version did one atomic per 8-row iteration and ran at 1.8 TB/s with 5 % of the rows passing.
//   2. the hit rows (arrival order is timing dependent) are radix-sorted by row, rescored exactly
//      (f64, index order -- rescore_rows_kernel), then STABLY radix-sorted by the f64 distance:
//      distance asc, row asc == the reference's stable sort over (document, line) order.
//   3. one D2H of the sorted pairs into pinned memory; the host cuts at the first distance that is not
//      < max_distance (exact f64 test) -- the guard-band extras sit behind the cut.
// Few hits (<= 2048) skip the device sorts: the host orders them.
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>

#include "common.h"
#include "device_utils.h"

namespace smt {

namespace {

struct ThrParams {
    const float *corpus;
    const float *query;
    uint64_t n_virtual;
    const uint64_t *chunk_table;  // FILTERED: row0 | valid rows << 32 per chunk (scan_kernels.hip)
    uint64_t n_chunks;
    float prefilter;
    uint32_t *hit_rows;
    unsigned long long *hit_count;
    uint64_t cap;
};

constexpr int HIT_BUF = 256;  // per-wave LDS buffer (u32 rows)

template <int U, bool FILTERED>
__global__ void __launch_bounds__(1024) scan_threshold_kernel(ThrParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_block = blockDim.x >> 6;
    volatile uint32_t *s_hits = reinterpret_cast<uint32_t *>(smem_raw) + wave * HIT_BUF;
    uint32_t *s_next = reinterpret_cast<uint32_t *>(smem_raw) + waves_per_block * HIT_BUF;

    const f32x4 q = reinterpret_cast<const f32x4 *>(p.query)[lane];
    const float a2 = wave_sum(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const bool qz = (a2 == 0.0f);
    const float rq = qz ? 0.0f : __frsqrt_rn(a2);

    const uint64_t n_chunks = p.n_chunks;
    const const_u64_ptr const_table = (const_u64_ptr)(uintptr_t)p.chunk_table;
    auto chunk_id = [&](uint32_t t) -> uint64_t {  // chunk t of this block (same deal as K2, see scan_kernels.hip)
        const uint64_t c = ((uint64_t)(t / waves_per_block) * gridDim.x + blockIdx.x) * waves_per_block + t % waves_per_block;
        return uniform_u64(c);  // the division runs on the VALU: tell the compiler the result is wave-uniform (scalar loads)
    };
    auto claim = [&]() -> uint32_t {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(s_next, 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    };
    auto fetch_desc = [&](uint64_t c) -> uint64_t {
        if (c >= n_chunks) return 0ull;
        if (FILTERED) return const_table[c];  // constant address space: scalar load (see scan_kernels.hip)
        const uint64_t v0 = c * U;
        const uint64_t left = p.n_virtual - v0;
        return v0 | ((left < (uint64_t)U ? left : (uint64_t)U) << 32);
    };
    auto issue_loads = [&](uint64_t desc, f32x4 (&c)[U], uint32_t (&row)[U]) {
        const uint32_t row0 = (uint32_t)desc, cnt = (uint32_t)(desc >> 32);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            row[j] = row0 + ((uint32_t)j < cnt ? (uint32_t)j : 0u);
            c[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p.corpus + (uint64_t)row[j] * 256) + lane);
        }
    };
    if (threadIdx.x == 0) *s_next = (uint32_t)waves_per_block;
    __syncthreads();

    uint32_t n_buf = 0;  // wave-uniform: hits waiting in s_hits
    auto flush = [&]() {
        if (n_buf == 0) return;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(p.hit_count, (unsigned long long)n_buf);
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
               (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base & 0xFFFFFFFFull));
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < n_buf; i += 64) {
            const unsigned long long slot = base + i;
            if (slot < p.cap) p.hit_rows[slot] = s_hits[i];  // beyond cap: counted, not stored (caller reruns)
        }
        __builtin_amdgcn_wave_barrier();
        n_buf = 0;
    };

    f32x4 cn[U];
    uint32_t rown[U];
    uint64_t cA = chunk_id((uint32_t)wave), cB = n_chunks, cC = n_chunks;
    uint64_t dA = fetch_desc(cA), dB = 0;
    if (cA < n_chunks) {
        issue_loads(dA, cn, rown);
        cB = chunk_id(claim());
        dB = fetch_desc(cB);
        if (FILTERED) cC = chunk_id(claim());  // only a table descriptor needs a stage of its own
    }
    while (cA < n_chunks) {
        f32x4 c[U];
        uint32_t row[U];
#pragma unroll
        for (int j = 0; j < U; ++j) { c[j] = cn[j]; row[j] = rown[j]; }
        const uint64_t first = dA & 0xFFFFFFFFull, end = first + (dA >> 32);  // rows [first, end) of this chunk are real
        if (cB < n_chunks) issue_loads(dB, cn, rown);
        uint64_t cN, dN;  // the chunk after B
        if (FILTERED) { cN = cC; dN = fetch_desc(cC); cC = chunk_id(claim()); }
        else { cN = chunk_id(claim()); dN = fetch_desc(cN); }

        // lane j (< U) remembers whether row j passed
        bool pass_mine = false;
        uint32_t my_row = 0;
        if constexpr (U == 4) {
            // the four rows of the chunk reduced together (device_utils.h wave_sum4: the sums of row j end in the lanes with
            // lane % 4 == j, i.e. lane j holds row j's), the distance computed lane-parallel: 2 x 15 + 1 x dist instead of
            // 8 x 11 + 4 x dist instructions per chunk, like the scan kernel
            float pb[4], pa[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pb[j] = c[j].x * c[j].x + c[j].y * c[j].y + c[j].z * c[j].z + c[j].w * c[j].w;
                pa[j] = c[j].x * q.x + c[j].y * q.y + c[j].z * q.z + c[j].w * q.w;
            }
            const float b2 = wave_sum4(pb[0], pb[1], pb[2], pb[3], lane);
            const float ab = wave_sum4(pa[0], pa[1], pa[2], pa[3], lane);
            const float d = dist_f32(ab, b2, rq, qz);
            const int jj = lane & 3;
            if (lane < 4) {
                pass_mine = (first + (uint64_t)jj < end) && (d < p.prefilter);
                my_row = jj == 0 ? row[0] : jj == 1 ? row[1] : jj == 2 ? row[2] : row[3];
            }
        } else
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const bool valid = first + j < end;  // written like K2's test: "j < count" makes clang branch around every row
            const float b2 = wave_sum(c[j].x * c[j].x + c[j].y * c[j].y + c[j].z * c[j].z + c[j].w * c[j].w);
            const float ab = wave_sum(c[j].x * q.x + c[j].y * q.y + c[j].z * q.z + c[j].w * q.w);
            const float d = dist_f32(ab, b2, rq, qz);
            if (lane == j) { pass_mine = valid && (d < p.prefilter); my_row = row[j]; }
        }
        const unsigned long long m = __ballot(pass_mine);
        if (m != 0ull) {
            const uint32_t cnt = (uint32_t)__popcll(m);
            if (n_buf + cnt > (uint32_t)HIT_BUF) flush();
            if (pass_mine) s_hits[n_buf + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = my_row;
            n_buf += cnt;
        }
        cA = cB; dA = dB;
        cB = cN; dB = dN;
    }
    flush();
}

__global__ void iota_bits_kernel(const double *dist, uint64_t *bits, uint64_t n)
{
    // distances are >= 0 (clipped) or NaN-free here, so the IEEE bit pattern orders like the value
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bits[i] = (uint64_t)__double_as_longlong(dist[i]);
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

int run_threshold_query(smt_ctx *ctx, const ThresholdQuery &t, const uint32_t **rows_host, const double **dist_host, uint64_t *n_pass)
{
    SMT_REQUIRE(t.rows < (1ull << 32), "a shard holds fewer than 2^32 rows");
    *rows_host = nullptr;
    *dist_host = nullptr;
    *n_pass = 0;
    int blocks = ctx->tune.scan_blocks > 0 ? ctx->tune.scan_blocks : ctx->num_cus;
    const int threads = ctx->tune.scan_threads;
    const size_t smem = (size_t)(threads / 64) * HIT_BUF * sizeof(uint32_t) + 16;
    const float prefilter = (float)(t.max_distance + 8e-6) + 0.0f;

    uint64_t cap = std::min<uint64_t>(t.n_virtual, (uint64_t)1 << 20);
    for (;;) {
        // scratch: rows_a | rows_b | dist_a | bits_a | bits_b | count | rocprim temp
        size_t temp_keys = 0, temp_pairs = 0;
        SMT_HIP_CHECK(rocprim::radix_sort_keys(nullptr, temp_keys, (uint32_t *)nullptr, (uint32_t *)nullptr, cap, 0, 32, ctx->stream));
        SMT_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, temp_pairs, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                                (uint32_t *)nullptr, cap, 0, 64, ctx->stream));
        const size_t b_rows = align_up((size_t)cap * sizeof(uint32_t), 256);
        const size_t b_f64 = align_up((size_t)cap * sizeof(double), 256);
        const size_t b_temp = align_up(std::max(temp_keys, temp_pairs), 256);
        const bool filtered = t.n_ranges > 0;
        const uint64_t n_chunks = filtered ? t.n_chunks : (t.n_virtual + 3) / 4;
        const size_t b_table = filtered ? align_up((size_t)n_chunks * sizeof(uint64_t), 256) : 0;
        int rc = ensure_scratch(ctx, 2 * b_rows + 3 * b_f64 + 256 + b_temp + b_table);
        if (rc) return rc;
        char *base = reinterpret_cast<char *>(ctx->d_scratch);
        uint32_t *rows_a = reinterpret_cast<uint32_t *>(base);
        uint32_t *rows_b = reinterpret_cast<uint32_t *>(base + b_rows);
        double *dist_a = reinterpret_cast<double *>(base + 2 * b_rows);
        uint64_t *bits_a = reinterpret_cast<uint64_t *>(base + 2 * b_rows + b_f64);
        uint64_t *bits_b = reinterpret_cast<uint64_t *>(base + 2 * b_rows + 2 * b_f64);
        unsigned long long *d_count = reinterpret_cast<unsigned long long *>(base + 2 * b_rows + 3 * b_f64);
        void *temp = base + 2 * b_rows + 3 * b_f64 + 256;
        uint64_t *table = reinterpret_cast<uint64_t *>(base + 2 * b_rows + 3 * b_f64 + 256 + b_temp);

        ThrParams p;
        p.corpus = t.corpus;
        p.query = t.query;
        p.n_virtual = t.n_virtual;
        p.chunk_table = filtered ? table : nullptr;
        p.n_chunks = n_chunks;
        p.prefilter = prefilter;
        p.hit_rows = rows_a;
        p.hit_count = d_count;
        p.cap = cap;
        SMT_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), ctx->stream));
        prof_begin(ctx, "scan");
        if (filtered && (rc = launch_build_chunk_table(ctx, t.ranges, t.range_chunk_prefix, t.n_ranges, n_chunks, table))) return rc;
        if (filtered) hipLaunchKernelGGL((scan_threshold_kernel<4, true>), dim3(blocks), dim3(threads), smem, ctx->stream, p);
        else hipLaunchKernelGGL((scan_threshold_kernel<4, false>), dim3(blocks), dim3(threads), smem, ctx->stream, p);
        prof_end(ctx, "scan");
        SMT_HIP_CHECK(hipGetLastError());

        rc = ensure_pinned(ctx, 64);
        if (rc) return rc;
        unsigned long long *h_count = reinterpret_cast<unsigned long long *>(ctx->h_pinned);
        SMT_HIP_CHECK(hipMemcpyAsync(h_count, d_count, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        const uint64_t n_hits = *h_count;
        if (n_hits > cap) { cap = n_hits; continue; }  // rare: rerun with an exact-size buffer
        if (n_hits == 0) return SMT_OK;

        const bool device_order = n_hits > 2048;
        const uint32_t *rows_final = rows_a;
        const double *dist_final = dist_a;
        if (device_order) {
            size_t tb = b_temp;
            SMT_HIP_CHECK(rocprim::radix_sort_keys(temp, tb, rows_a, rows_b, n_hits, 0, 32, ctx->stream));
            if ((rc = launch_rescore_rows(ctx, t.corpus, t.query, rows_b, n_hits, dist_a))) return rc;
            hipLaunchKernelGGL(iota_bits_kernel, dim3((unsigned)((n_hits + 255) / 256)), dim3(256), 0, ctx->stream, dist_a, bits_a, n_hits);
            tb = b_temp;
            SMT_HIP_CHECK(rocprim::radix_sort_pairs(temp, tb, bits_a, bits_b, rows_b, rows_a, n_hits, 0, 64, ctx->stream));
            rows_final = rows_a;                                      // sorted by (distance bits, row)
            dist_final = reinterpret_cast<const double *>(bits_b);    // the sorted bit patterns ARE the distances
        } else {
            if ((rc = launch_rescore_rows(ctx, t.corpus, t.query, rows_a, n_hits, dist_a))) return rc;
        }
        const size_t h_rows_bytes = align_up((size_t)n_hits * sizeof(uint32_t), 64);
        rc = ensure_pinned(ctx, h_rows_bytes + (size_t)n_hits * sizeof(double) + 64);
        if (rc) return rc;
        uint32_t *h_rows = reinterpret_cast<uint32_t *>(ctx->h_pinned);
        double *h_dist = reinterpret_cast<double *>(reinterpret_cast<char *>(ctx->h_pinned) + h_rows_bytes);
        SMT_HIP_CHECK(hipMemcpyAsync(h_rows, rows_final, (size_t)n_hits * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        SMT_HIP_CHECK(hipMemcpyAsync(h_dist, dist_final, (size_t)n_hits * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));

        uint64_t n_ok;
        if (device_order) {
            n_ok = (uint64_t)(std::lower_bound(h_dist, h_dist + n_hits, t.max_distance) - h_dist);  // strict <
        } else {
            // few hits: exact strict test, then the reference's order (distance asc, row asc) on the host
            std::vector<std::pair<double, uint32_t>> v;
            v.reserve(n_hits);
            for (uint64_t i = 0; i < n_hits; ++i) if (h_dist[i] < t.max_distance) v.emplace_back(h_dist[i], h_rows[i]);
            std::sort(v.begin(), v.end());
            for (size_t i = 0; i < v.size(); ++i) { h_dist[i] = v[i].first; h_rows[i] = v[i].second; }
            n_ok = v.size();
        }
        *rows_host = h_rows;
        *dist_host = h_dist;
        *n_pass = n_ok;
        return SMT_OK;
    }
}

}  // namespace smt
