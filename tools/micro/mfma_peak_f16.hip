// What does the 16-bit MFMA pipe of THIS part sustain?  v_mfma_f32_32x32x16_f16 back to back, ~4 ms per launch (the length of
// a K3 main level), with the operand sources K3 has:
//   "reg"      : both operands in registers, one dependent accumulator chain per wave (K3's shape: 16 MFMAs into one f32x16)
//   "reg2"     : two independent chains per wave
//   "lds"      : the B operand of every MFMA read from LDS (ds_read_b128, groups of four double-buffered as in K3's f16 x 1 sweep)
//   "lds_epi"  : "lds" + K3's epilogue shape after every 16 MFMAs (wait for the accumulator, a 16-way max, one compare, re-zero)
// each with 1 and 2 waves per SIMD (256 / 512 threads per CU).  Per launch: wall time (HIP events), issued PFLOP/s, and the shader
// clock the kernel ran at = s_memtime ticks / s_memrealtime (100 MHz) ticks, taken inside the kernel by wave 0 of every block.
// Operands: small constants (few bits toggle between MFMAs) or random fp16 in [-1, 1) as a unit-vector corpus gives them -- the
// part is power-managed, and what the pipe sustains depends on the data it multiplies.
// Output: one JSON object per line.   Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak_f16.hip -o tools/micro/mfma_peak_f16
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t rnd_f16x2(uint32_t x)   // two fp16 in [-1, 1) with random mantissas
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    // sign random, exponent in {0x38..0x3b} (0.5 .. 1), mantissa random
    return (x & 0x83ff83ffu) | 0x38003800u | ((x >> 3) & 0x04000400u);
}
template <int VARIANT, bool RANDOM>
__global__ void __launch_bounds__(512) mfma_loop(float *out, unsigned long long *clocks, int iters, uint32_t seed)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    u32x4 *s_b = reinterpret_cast<u32x4 *>(smem);          // 32 query rows x 33 quads (K3's padded 512-B rows)
    for (int i = threadIdx.x; i < 32 * 33; i += blockDim.x) {
        if (RANDOM) s_b[i] = (u32x4){rnd_f16x2(4 * i + seed), rnd_f16x2(4 * i + 1 + seed), rnd_f16x2(4 * i + 2 + seed), rnd_f16x2(4 * i + 3 + seed)};
        else s_b[i] = (u32x4){0x3c003c00u ^ seed, 0x38003800u, 0x34003400u, 0x30003000u};
    }
    __syncthreads();
    u32x4 A[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const uint32_t b = (uint32_t)(blockIdx.x * 512 + threadIdx.x) * 64 + m * 4 + seed * 7919u;
        if (RANDOM) A[m] = (u32x4){rnd_f16x2(b), rnd_f16x2(b + 1), rnd_f16x2(b + 2), rnd_f16x2(b + 3)};
        else A[m] = (u32x4){0x3c003c00u + (uint32_t)m, 0x38003800u ^ seed, 0x34003400u, 0x30003000u + (uint32_t)lane};
    }
    const u32x4 *bq = s_b + (lane & 31) * 33 + (lane >> 5);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float keep = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if constexpr (VARIANT == 0) {
#pragma unroll
            for (int m = 0; m < 16; ++m)
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[m]), __builtin_bit_cast(f16x8, A[(m + 1) & 15]), acc0, 0, 0, 0);
        } else if constexpr (VARIANT == 1) {
#pragma unroll
            for (int m = 0; m < 16; m += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[m]), __builtin_bit_cast(f16x8, A[m + 1]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[m + 1]), __builtin_bit_cast(f16x8, A[m]), acc1, 0, 0, 0);
            }
        } else {
            u32x4 B[2][4];
#pragma unroll
            for (int d = 0; d < 4; ++d) B[0][d] = bq[2 * d];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                if (g + 1 < 4) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) B[(g + 1) & 1][d] = bq[2 * (4 * (g + 1) + d)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[4 * g + d]), __builtin_bit_cast(f16x8, B[g & 1][d]), acc0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (VARIANT == 3) {
                float mx = acc0[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc0[r]);
                if (mx >= 1e30f) keep += mx;      // never: products are small
#pragma unroll
                for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = keep;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int VARIANT, bool RANDOM>
static void run(const char *name, int threads, int cus, float *out, unsigned long long *clocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<unsigned long long> h(2 * cus);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_loop<VARIANT, RANDOM>), dim3(cus), dim3(threads), 32 * 33 * 16, 0, out, clocks, iters, (uint32_t)rep);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost);
        double ratio = 0;
        for (int b = 0; b < cus; ++b) ratio += (double)h[2 * b] / (double)h[2 * b + 1];
        ratio /= cus;
        const double flops = (double)cus * (threads / 64) * iters * 16.0 * 32768.0;
        printf("{\"variant\": \"%s\", \"operands\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"issued_PFLOPs\": %.3f, \"frac_of_2.5PF\": %.3f, "
               "\"cyclecounter_ticks_per_100MHz_tick\": %.3f, \"mfma_cycles_over_cyclecounter\": %.3f}\n",
               name, RANDOM ? "random fp16" : "constants", threads / 256, ms, flops / (ms * 1e-3) / 1e15, flops / (ms * 1e-3) / 2.5e15, ratio,
               (double)iters * 16.0 * 32.0 * (threads / 256) / ((double)h[0]));
        fflush(stdout);
    }
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *out;
    unsigned long long *clocks;
    hipMalloc(&out, (size_t)cus * 512 * 4);
    hipMalloc(&clocks, (size_t)cus * 16);
    const int iters = 12000;   // 2 waves/SIMD: 12000 x 16 x 32 cycles x 2 = 12.3 M cycles ~ 5-7 ms
    for (int threads : {256, 512}) {
        run<0, false>("reg", threads, cus, out, clocks, iters);
        run<1, false>("reg2", threads, cus, out, clocks, iters);
        run<2, false>("lds", threads, cus, out, clocks, iters);
        run<3, false>("lds_epi", threads, cus, out, clocks, iters);
        run<1, true>("reg2", threads, cus, out, clocks, iters);
        run<2, true>("lds", threads, cus, out, clocks, iters);
        run<3, true>("lds_epi", threads, cus, out, clocks, iters);
    }
    return 0;
}
