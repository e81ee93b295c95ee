// wave_sum4 (csrc/device_utils.h) against a host sum: lane l must hold the 64-lane sum of input l % 4, every lane, for integer-valued
// floats (exact in any order) -- the DPP rotations and v_permlane16_swap / v_permlane32_swap mean what the helper assumes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../semtools_amd/csrc/device_utils.h"
__global__ void k(const float *in, float *out)
{
    const int lane = threadIdx.x & 63;
    out[threadIdx.x] = smt::wave_sum4(in[lane], in[64 + lane], in[128 + lane], in[192 + lane], lane);
}
int main()
{
    float h[256], want[4] = {0, 0, 0, 0}, got[64];
    for (int i = 0; i < 256; ++i) { h[i] = (float)((i * 7919) % 1013 - 500); want[i / 64] += h[i]; }
    float *d_in, *d_out;
    hipMalloc(&d_in, sizeof(h)); hipMalloc(&d_out, sizeof(got));
    hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out);
    hipMemcpy(got, d_out, sizeof(got), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) if (got[l] != want[l % 4]) { if (bad < 8) printf("lane %d: got %g want %g\n", l, got[l], want[l % 4]); ++bad; }
    printf(bad ? "FAIL (%d lanes)\n" : "PASS wave_sum4\n", bad);
    return bad != 0;
}
