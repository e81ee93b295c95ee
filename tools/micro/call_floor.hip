// call_floor.hip -- what does ONE synchronous host call cost on this box before it does any work?  (VERDICT r5 "next" 6: small
// calls are launch-bound -- a 1-query search over 4096 rows is 43 us.)  Every figure is the median of 7 rounds of 300 calls, us.
//   sync_only            hipStreamSynchronize on an idle stream
//   kernel+sync          one empty kernel, then hipStreamSynchronize
//   2kernels+sync        two dependent empty kernels, then sync (the scan -> select hand-over)
//   h2d+kernel+d2h+sync  1 KiB up (pinned), empty kernel, 256 B down (pinned), sync: today's small host-form call without its work
//   kernel+flag          one kernel that writes a sequence number to PINNED host memory with a system-scope release; the host spins on
//                        that word instead of asking the runtime -- no signal wait, no interrupt, no runtime bookkeeping on the way back
//   kernel(rd pinned)+flag  the same, the kernel also READS 1 KiB of pinned host memory first (the query where the caller left it)
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/call_floor.hip -o tools/micro/call_floor
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 1024) *p = 1; }
__global__ void flag_kernel(const float *q_pinned, float *sink, unsigned long long *flag, unsigned long long seq)
{
    if (q_pinned) {
        float v = q_pinned[threadIdx.x];
        if (v == 123.456f) *sink = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename F>
static double median_us(F &&f, int reps = 300, int rounds = 7)
{
    for (int i = 0; i < 50; ++i) f();
    std::vector<double> v;
    for (int r = 0; r < rounds; ++r) {
        const double t0 = now_us();
        for (int i = 0; i < reps; ++i) f();
        v.push_back((now_us() - t0) / reps);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main()
{
    hipStream_t st;
    CHECK(hipSetDevice(0));
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *h_in, *d_buf, *h_out;
    unsigned long long *h_flag;
    CHECK(hipHostMalloc(&h_in, 4096, hipHostMallocDefault));
    CHECK(hipHostMalloc(&h_out, 4096, hipHostMallocDefault));
    CHECK(hipHostMalloc(&h_flag, 64, hipHostMallocDefault));
    CHECK(hipMalloc(&d_buf, 1 << 20));
    for (int i = 0; i < 1024; ++i) h_in[i] = (float)i;
    *h_flag = 0;
    unsigned long long seq = 0;
    volatile unsigned long long *vf = h_flag;

    const double a = median_us([&] { (void)hipStreamSynchronize(st); });
    const double b = median_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, nullptr); (void)hipStreamSynchronize(st); });
    const double c = median_us([&] {
        hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, st, nullptr);
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(1024), 0, st, nullptr);
        (void)hipStreamSynchronize(st);
    });
    const double d = median_us([&] {
        (void)hipMemcpyAsync(d_buf, h_in, 1024, hipMemcpyHostToDevice, st);
        hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, st, nullptr);
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(1024), 0, st, nullptr);
        (void)hipMemcpyAsync(h_out, d_buf + 4096, 256, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
    });
    const double e = median_us([&] {
        ++seq;
        hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(256), 0, st, nullptr, d_buf, h_flag, seq);
        while (*vf != seq) {}
    });
    const double f = median_us([&] {
        ++seq;
        hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(256), 0, st, h_in, d_buf, h_flag, seq);
        while (*vf != seq) {}
    });
    // ... and 16 calls in a row with the spin, then ONE runtime sync (the runtime still has to retire its signals some time)
    const double g = median_us([&] {
        for (int i = 0; i < 16; ++i) {
            ++seq;
            hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(256), 0, st, h_in, d_buf, h_flag, seq);
            while (*vf != seq) {}
        }
        (void)hipStreamSynchronize(st);
    }, 40) / 16;
    printf("{\"sync_only_us\": %.2f, \"kernel_sync_us\": %.2f, \"two_kernels_sync_us\": %.2f, \"h2d_2kernels_d2h_sync_us\": %.2f, "
           "\"kernel_flag_spin_us\": %.2f, \"kernel_reads_pinned_flag_spin_us\": %.2f, \"flag_spin_x16_then_sync_us_per_call\": %.2f}\n",
           a, b, c, d, e, f, g);
    return 0;
}
