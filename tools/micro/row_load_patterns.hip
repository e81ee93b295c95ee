// How fast do 32-row x 1 KiB tiles arrive in REGISTERS on MI355X, as a function of how a wave's 32 load instructions
// cut the tile?  (K3 keeps a row tile in registers for a sweep over the query tiles; the MFMA operand layout wants
// lane j <-> row j, i.e. 16-B pieces of 32 different rows per instruction.)  8 waves per CU, every wave: 32 loads in
// flight, then NMFMA dependent bf16 MFMAs (the sweep), repeat.  Prints TB/s per pattern and sweep length.
//   pattern 0: lane (j,h) float4 index 4m+2h, 4m+2h+1      16-B pieces, two per 64 B        (bf16 x 3 layout)
//   pattern 1: lane (j,h) float4 index 2i+h                32-B runs per row per instruction (f32 MFMA layout)
//   pattern 2: instr (s,u): row 8u + l/8, float4 8s + l%8  128-B runs (would need an LDS transpose afterwards)
//   pattern 3: instr i: row i, float4 l                    whole rows (1 KiB runs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PATTERN, int NMFMA>
__global__ void __launch_bounds__(512, 2) tiles(const float *corpus, unsigned long long n_tiles, float *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, j = lane & 31;
    const unsigned long long W = (unsigned long long)gridDim.x * 8;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float sum = 0.f;
    for (unsigned long long t = (unsigned long long)blockIdx.x * 8 + wave; t < n_tiles; t += W) {
        const f32x4 *base = reinterpret_cast<const f32x4 *>(corpus + t * 32 * 256);
        f32x4 R[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const f32x4 *p;
            if (PATTERN == 0) p = base + j * 64 + 4 * (i >> 1) + 2 * h + (i & 1);
            else if (PATTERN == 1) p = base + j * 64 + 2 * i + h;
            else if (PATTERN == 2) p = base + (8 * (i & 3) + (lane >> 3)) * 64 + 8 * (i >> 2) + (lane & 7);
            else p = base + i * 64 + lane;
            R[i] = __builtin_nontemporal_load(p);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += R[i].x + R[i].y + R[i].z + R[i].w;
        sum += s;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)s; b[e] = (__bf16)1.0f; }
#pragma unroll 8
        for (int m = 0; m < NMFMA; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    float o = sum;
    for (int r = 0; r < 16; ++r) o += acc[r];
    out[blockIdx.x * 512 + threadIdx.x] = o;
}

__global__ void fill_random(float *d, unsigned long long n)
{
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long x = i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        d[i] = (float)(int)(x & 0xFFFFF) * (1.0f / 1048576.0f) - 0.5f;
    }
}

template <int PATTERN, int NMFMA>
static void run(const float *d, unsigned long long n_tiles, float *out, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((tiles<PATTERN, NMFMA>), dim3(blocks), dim3(512), 0, 0, d, n_tiles, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("{\"pattern\": %d, \"mfma_per_tile\": %d, \"ms\": %.3f, \"TBps\": %.2f}\n", PATTERN, NMFMA, best,
           (double)n_tiles * 32768.0 / (best * 1e-3) / 1e12);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const unsigned long long n_tiles = 312500;  // 10 M rows
    float *d, *out;
    hipMalloc(&d, n_tiles * 32768);
    // random contents: an all-zero buffer reads ~5 % faster than real data (less DRAM / fabric power)
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, d, n_tiles * 8192ull);
    hipDeviceSynchronize();
    hipMalloc(&out, (size_t)prop.multiProcessorCount * 512 * 4);
    const int blocks = prop.multiProcessorCount;
    run<0, 0>(d, n_tiles, out, blocks);   run<1, 0>(d, n_tiles, out, blocks);   run<2, 0>(d, n_tiles, out, blocks);   run<3, 0>(d, n_tiles, out, blocks);
    run<0, 96>(d, n_tiles, out, blocks);  run<1, 96>(d, n_tiles, out, blocks);  run<2, 96>(d, n_tiles, out, blocks);  run<3, 96>(d, n_tiles, out, blocks);
    run<0, 192>(d, n_tiles, out, blocks); run<1, 192>(d, n_tiles, out, blocks); run<2, 192>(d, n_tiles, out, blocks); run<3, 192>(d, n_tiles, out, blocks);
    return 0;
}
