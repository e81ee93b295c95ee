// gather_patterns.hip -- what do RANDOM 1 KiB row gathers get out of an MI355X?  (VERDICT r2 "next" 7: K1, the embedding
// gather, sits at 0.66 of HBM on uniform ids; is that the kernel or the part?)
//
// A 4 GB table of 1 KiB rows (4 M rows: 16 x the 256 MB MALL, no cache help), 32 M gathers (32 GB) per run, rows summed so
// that nothing is optimised away.  Variants:
//   shape   "g16"  : 16 lanes per row, 4 instructions x 256 contiguous bytes (K1's mapping: 4 rows per wave instruction)
//           "wave" : 64 lanes per row, one instruction = the whole 1 KiB row
//   depth   rows in flight per lane group (g16) / per wave (wave): 2, 4, 8
//   ids     "hash" : the row number is computed in registers (no dependent load)
//           "mem"  : the row number is LOADED from an id array first (K1: ids -> row address -> row), the two memory
//                    latencies in series
//   span    1 KiB rows, or runs of 2 / 4 consecutive rows (2 / 4 KiB contiguous): how much of the loss is DRAM page locality
// plus a plain streaming read of the same bytes as the ceiling.  Output: one JSON object per line.
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/gather_patterns.hip -o tools/micro/gather_patterns
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (uint32_t)x;
}

// SHAPE 0: g16, 1: wave.  IDS 0: hash, 1: mem.  Each lane group (16 lanes / 64 lanes) walks `per_group` rows.
template <int SHAPE, int DEPTH, int IDS>
__global__ void __launch_bounds__(256) gather_kernel(const f32x4 *table, uint32_t row_mask, uint32_t span, const uint32_t *ids,
                                                     uint64_t per_group, float *sink)
{
    const int lane = threadIdx.x & 63;
    constexpr int GL = SHAPE == 0 ? 16 : 64;                 // lanes per group
    const int a = lane & (GL - 1);
    const uint64_t group = (((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / GL);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (uint64_t t = 0; t < per_group; t += DEPTH) {
        f32x4 r[DEPTH][SHAPE == 0 ? 4 : 1];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const uint64_t i = group * per_group + t + u;
            // runs of `span` consecutive rows: the run start is random, the position inside the run cycles
            uint32_t row = IDS ? ids[i / span] : mix(i / span);
            row = ((row * span) + (uint32_t)(i % span)) & row_mask;
            const f32x4 *src = table + (size_t)row * 64;
            if (SHAPE == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) r[u][c] = __builtin_nontemporal_load(src + c * 16 + a);
            } else {
                r[u][0] = __builtin_nontemporal_load(src + a);
            }
        }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
#pragma unroll
            for (int c = 0; c < (SHAPE == 0 ? 4 : 1); ++c) acc += r[u][c];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;   // never true: keeps the loads alive
}

__global__ void __launch_bounds__(256) stream_kernel(const f32x4 *table, uint64_t n_f4, float *sink)
{
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < n_f4; i += 4 * stride) {
        const f32x4 v0 = __builtin_nontemporal_load(table + i), v1 = __builtin_nontemporal_load(table + i + stride);
        const f32x4 v2 = __builtin_nontemporal_load(table + i + 2 * stride), v3 = __builtin_nontemporal_load(table + i + 3 * stride);
        acc += v0 + v1 + v2 + v3;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <int SHAPE, int DEPTH, int IDS>
static int run(const char *shape, const char *ids_name, const f32x4 *table, uint32_t row_mask, uint32_t span, const uint32_t *ids,
               uint64_t total_rows, int waves_per_cu, int n_cu, float *sink)
{
    constexpr int GL = SHAPE == 0 ? 16 : 64;
    const uint64_t groups = (uint64_t)n_cu * waves_per_cu * (64 / GL);
    const uint64_t per_group = total_rows / groups / DEPTH * DEPTH;
    const unsigned blocks = (unsigned)(groups * GL / 256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((gather_kernel<SHAPE, DEPTH, IDS>), dim3(blocks), dim3(256), 0, 0, table, row_mask, span, ids, per_group, sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)groups * per_group * 1024.0;
    printf("{\"shape\": \"%s\", \"rows_in_flight_per_group\": %d, \"ids\": \"%s\", \"contiguous_bytes\": %u, \"waves_per_cu\": %d, "
           "\"ms\": %.3f, \"GBps\": %.0f, \"frac_of_8TBps\": %.3f}\n",
           shape, DEPTH, ids_name, span * 1024u, waves_per_cu, best, bytes / best / 1e6, bytes / best / 1e6 / 8000.0);
    fflush(stdout);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return 0;
}

int main()
{
    const uint64_t table_rows = 1ull << 22;            // 4 M rows x 1 KiB = 4 GiB
    const uint64_t total = 1ull << 25;                 // 32 M gathers = 32 GiB per run
    f32x4 *table;
    uint32_t *ids;
    float *sink;
    CHECK(hipMalloc(&table, table_rows * 1024));
    CHECK(hipMemset(table, 0, table_rows * 1024));
    CHECK(hipMalloc(&sink, 64));
    std::vector<uint32_t> h(total);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (auto &v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (uint32_t)(s >> 33); }
    CHECK(hipMalloc(&ids, total * 4));
    CHECK(hipMemcpy(ids, h.data(), total * 4, hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const uint32_t mask = (uint32_t)table_rows - 1;
    {   // ceiling: streaming read of the table, 8 times
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(stream_kernel, dim3(n_cu * 8), dim3(256), 0, 0, table, table_rows * 64, sink);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("{\"shape\": \"stream\", \"ms\": %.3f, \"GBps\": %.0f, \"frac_of_8TBps\": %.3f}\n", best, table_rows * 1024.0 / best / 1e6,
               table_rows * 1024.0 / best / 1e6 / 8000.0);
    }
    for (int wpc : {8, 16, 32}) {
        if (run<0, 2, 0>("g16", "hash", table, mask, 1, ids, total, wpc, n_cu, sink)) return 1;
        if (run<0, 4, 0>("g16", "hash", table, mask, 1, ids, total, wpc, n_cu, sink)) return 1;
        if (run<0, 8, 0>("g16", "hash", table, mask, 1, ids, total, wpc, n_cu, sink)) return 1;
        if (run<1, 2, 0>("wave", "hash", table, mask, 1, ids, total, wpc, n_cu, sink)) return 1;
        if (run<1, 4, 0>("wave", "hash", table, mask, 1, ids, total, wpc, n_cu, sink)) return 1;
        if (run<1, 8, 0>("wave", "hash", table, mask, 1, ids, total, wpc, n_cu, sink)) return 1;
    }
    // the dependent id load (K1's real pattern) at the best occupancy
    if (run<0, 4, 1>("g16", "mem", table, mask, 1, ids, total, 16, n_cu, sink)) return 1;
    if (run<0, 8, 1>("g16", "mem", table, mask, 1, ids, total, 16, n_cu, sink)) return 1;
    if (run<1, 8, 1>("wave", "mem", table, mask, 1, ids, total, 16, n_cu, sink)) return 1;
    if (run<0, 4, 1>("g16", "mem", table, mask, 1, ids, total, 32, n_cu, sink)) return 1;
    // longer contiguous runs: DRAM page locality
    for (uint32_t span : {2u, 4u, 16u}) {
        if (run<1, 8, 0>("wave", "hash", table, mask, span, ids, total, 16, n_cu, sink)) return 1;
        if (run<0, 4, 0>("g16", "hash", table, mask, span, ids, total, 16, n_cu, sink)) return 1;
    }
    return 0;
}
