// Achievable f32 MFMA rate on this part: v_mfma_f32_32x32x2_f32 back to back, no memory traffic.
// 2 waves per SIMD (like K3), two independent accumulator chains per wave.  Prints TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(512, 2) mfma_loop(float *out, int iters, float a0, float b0)
{
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus, threads = 512, iters = 20000;
    float *out;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * (threads / 64) * iters * 32.0 * 4096.0;
        printf("{\"cus\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f, \"clock_MHz_reported\": %d}\n", cus, ms, flops / (ms * 1e-3) / 1e12, prop.clockRate / 1000);
    }
    return 0;
}
