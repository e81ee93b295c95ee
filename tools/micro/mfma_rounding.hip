// mfma_rounding.hip -- how do the gfx950 MFMAs round?  (VERDICT r2 "weak" 3: the certificate's error bounds doubled the
// accumulation term "because the MFMA adder tree is not documented to round to nearest".)  This probe measures it.
//
// Every test computes ONE scalar  d = c + sum_k a_k * b_k  with the instruction under test (all rows of A equal, all
// columns of B equal, so every output element is that scalar) with operands chosen so that the result tells the rounding
// apart:  ulp = 2^-23 * 1.0 is the spacing of f32 in [1, 2); c = 1.5 sits in the middle of that binade.
//   single product 0.75 ulp        -> nearest: c + ulp      toward zero / -inf: c
//   single product 0.25 ulp        -> nearest: c            toward +inf: c + ulp
//   tie 0.5 ulp on even / odd c    -> nearest-even: c / c + 2 ulp
//   K products summing to 0.75 ulp, each < 0.5 ulp -> one rounding of the exact sum: c + ulp; a chain of rounded adds: c
//   a product needing > 24 bits that cancels against another -> products kept exact inside the instruction or rounded first
//   negative mirror images
// plus an empirical part: 256-dim dot products of positive values through chains of MFMAs exactly as the kernels issue them
// (f32: 128 x 32x32x2; bf16 / f16: 16 x 32x32x16), max |result - exact| in units of the exact value * 2^-24.
//
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_rounding.hip -o tools/micro/mfma_rounding
// Output: one JSON object on stdout (profiles/r03_mfma_rounding.json holds a run).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// one wave; a[k], b[k] for k < K: every A row / B column is the same K-vector.  kind 0: f32 32x32x2 (K = 2 per
// instruction, n_steps instructions chained through the accumulator), 1: bf16 32x32x16, 2: f16 32x32x16 (K = 16).
__global__ void probe_kernel(int kind, int n_steps, const float *a, const float *b, float c, float *out)
{
    const int lane = threadIdx.x, h = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c;
    for (int s = 0; s < n_steps; ++s) {
        if (kind == 0) {
            // lane (i, h) feeds A[i][k = h] and B[k = h][j]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s + h], b[2 * s + h], acc, 0, 0, 0);
        } else if (kind == 1) {
            bf16x8 av, bv;   // lane (j, h) feeds dims 8h .. 8h + 7 of the 16
            for (int e = 0; e < 8; ++e) { av[e] = (__bf16)a[16 * s + 8 * h + e]; bv[e] = (__bf16)b[16 * s + 8 * h + e]; }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
        } else {
            f16x8 av, bv;
            for (int e = 0; e < 8; ++e) { av[e] = (_Float16)a[16 * s + 8 * h + e]; bv[e] = (_Float16)b[16 * s + 8 * h + e]; }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
        }
    }
    if (lane == 0) out[0] = acc[0];
    if (lane == 37) out[1] = acc[5];   // another element of the tile: must be the same scalar
}

static float *d_a, *d_b, *d_out;

static int run(int kind, const std::vector<float> &a, const std::vector<float> &b, float c, float *result)
{
    const int K = kind == 0 ? 2 : 16;
    const int n_steps = (int)a.size() / K;
    CHECK(hipMemcpy(d_a, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_b, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, kind, n_steps, d_a, d_b, c, d_out);
    CHECK(hipGetLastError());
    float o[2];
    CHECK(hipMemcpy(o, d_out, 8, hipMemcpyDeviceToHost));
    if (memcmp(&o[0], &o[1], 4) != 0) fprintf(stderr, "warning: tile elements differ (%a vs %a)\n", o[0], o[1]);
    *result = o[0];
    return 0;
}

static const char *KIND[3] = {"v_mfma_f32_32x32x2_f32", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16"};

int main()
{
    CHECK(hipMalloc(&d_a, 4096 * 4));
    CHECK(hipMalloc(&d_b, 4096 * 4));
    CHECK(hipMalloc(&d_out, 16));
    const float ulp = ldexpf(1.0f, -23), c = 1.5f;
    printf("{\n");
    for (int kind = 0; kind < 3; ++kind) {
        const int K = kind == 0 ? 2 : 16;
        auto single = [&](float p_over_ulp, float cc, float *res) {   // one product p = p_over_ulp * ulp, the rest zero
            std::vector<float> a(K, 0.0f), b(K, 0.0f);
            a[0] = p_over_ulp;   // exactly representable in bf16 / f16 for the values used (0.75, 0.25, 0.5, ...)
            b[0] = ulp;          // 2^-23: a power of two (f16: subnormal range starts at 2^-14 -> use scaling below)
            if (kind == 2) { a[0] = p_over_ulp * ldexpf(1.0f, -10); b[0] = ldexpf(1.0f, -13); }   // 2^-10 * 2^-13 = 2^-23, both normal in fp16
            return run(kind, a, b, cc, res);
        };
        float r;
        printf("  \"%s\": {\n", KIND[kind]);
        if (single(0.75f, c, &r)) return 1;
        const bool up_075 = r == c + ulp;
        printf("    \"c_plus_0.75ulp\": \"%a\", \"rounds_up_at_0.75ulp\": %s,\n", r, up_075 ? "true" : "false");
        if (single(0.25f, c, &r)) return 1;
        const bool up_025 = r == c + ulp;
        printf("    \"c_plus_0.25ulp\": \"%a\", \"rounds_up_at_0.25ulp\": %s,\n", r, up_025 ? "true" : "false");
        if (single(-0.75f, c, &r)) return 1;
        const bool down_075 = r == c - ulp;
        printf("    \"c_minus_0.75ulp\": \"%a\", \"rounds_down_at_-0.75ulp\": %s,\n", r, down_075 ? "true" : "false");
        if (single(-0.25f, c, &r)) return 1;
        const bool down_025 = r == c - ulp;
        printf("    \"c_minus_0.25ulp\": \"%a\", \"rounds_down_at_-0.25ulp\": %s,\n", r, down_025 ? "true" : "false");
        float tie_even, tie_odd;
        if (single(0.5f, c, &tie_even)) return 1;            // c = 1.5: mantissa even -> nearest-even keeps c
        if (single(0.5f, c + ulp, &tie_odd)) return 1;       // odd mantissa -> nearest-even goes to c + 2 ulp
        const bool ties_even = tie_even == c && tie_odd == c + 2 * ulp;
        printf("    \"tie_on_even\": \"%a\", \"tie_on_odd\": \"%a\", \"ties_to_even\": %s,\n", tie_even, tie_odd, ties_even ? "true" : "false");
        // K products, each well below half an ulp, summing to 0.75 ulp
        {
            std::vector<float> a(K), b(K);
            const float each = 0.75f / K;                    // 0.375 (K = 2) or 0.046875 = 3 * 2^-6 (K = 16): exact in bf16 / f16
            for (int k = 0; k < K; ++k) { a[k] = each; b[k] = ulp; if (kind == 2) { a[k] = each * ldexpf(1.0f, -10); b[k] = ldexpf(1.0f, -13); } }
            if (run(kind, a, b, c, &r)) return 1;
            printf("    \"k_small_products_sum_0.75ulp\": \"%a\", \"products_of_one_instruction_summed_before_rounding\": %s,\n", r,
                   r == c + ulp ? "true" : "false");
        }
        // the same 0.75 ulp spread over TWO chained instructions (0.375 ulp each): each instruction rounds its own result
        {
            std::vector<float> a(2 * K, 0.0f), b(2 * K, 0.0f);
            for (int s = 0; s < 2; ++s) { a[s * K] = 0.375f; b[s * K] = ulp; if (kind == 2) { a[s * K] = 0.375f * ldexpf(1.0f, -10); b[s * K] = ldexpf(1.0f, -13); } }
            if (run(kind, a, b, c, &r)) return 1;
            printf("    \"two_chained_instructions_0.375ulp_each\": \"%a\", \"accumulator_rounded_after_every_instruction\": %s,\n", r,
                   r == c ? "true" : "false");
        }
        // are products kept exact?  p0 = (1 + 2^-e)^2 = 1 + 2^-(e-1) + 2^-2e, p1 = -(1 + 2^-(e-1)), c = 0: exact result 2^-2e
        {
            std::vector<float> a(K, 0.0f), b(K, 0.0f);
            const int e = kind == 0 ? 13 : kind == 1 ? 7 : 10;   // operands with 14 / 8 / 11 significant bits
            a[0] = b[0] = 1.0f + ldexpf(1.0f, -e);
            a[1] = -(1.0f + ldexpf(1.0f, -(e - 1)));
            b[1] = 1.0f;
            if (run(kind, a, b, 0.0f, &r)) return 1;
            printf("    \"cancelling_products\": \"%a\", \"expected_if_products_exact\": \"%a\", \"products_exact\": %s,\n", r, ldexpf(1.0f, -2 * e),
                   r == ldexpf(1.0f, -2 * e) ? "true" : "false");
        }
        // a small product next to a large accumulator and a large product: is the small one lost before the sum?
        {
            std::vector<float> a(K, 0.0f), b(K, 0.0f);
            a[0] = 1.0f; b[0] = 1.0f;                          // + 1
            a[1] = -1.0f; b[1] = 1.0f;                         // - 1
            if (K > 2) { a[2] = 0.75f; b[2] = ulp; if (kind == 2) { a[2] = 0.75f * ldexpf(1.0f, -10); b[2] = ldexpf(1.0f, -13); } }
            if (K > 2) {
                if (run(kind, a, b, c, &r)) return 1;
                printf("    \"plus1_minus1_plus_0.75ulp\": \"%a\", \"small_product_survives_large_cancelling_pair\": %s,\n", r, r == c + ulp ? "true" : "false");
            }
        }
        // How far below the accumulator's ulp does the instruction still SEE a product?  p0 = 0.5 ulp (a tie: stays at c, even
        // mantissa) + p1 = 2^-j ulp: seen -> above the tie -> c + ulp; dropped -> c.  Negative twin on an odd accumulator
        // (tie -> c + 2 ulp if p1 is dropped, c + ulp if seen).  The largest j that is still seen = guard bits of the adder.
        {
            int seen_pos = 0, seen_neg = 0;
            for (int j = 1; j <= 40; ++j) {
                std::vector<float> a(K, 0.0f), b(K, 0.0f);
                a[0] = 0.5f; b[0] = ulp;
                a[1] = 1.0f; b[1] = ldexpf(ulp, -j);
                if (kind == 2) {   // keep both fp16 operands normal: split the exponent over a and b
                    a[0] = 0.5f * ldexpf(1.0f, -10); b[0] = ldexpf(1.0f, -13);
                    a[1] = ldexpf(1.0f, -12 - j / 2); b[1] = ldexpf(1.0f, -11 - (j - j / 2));
                    if (-12 - j / 2 < -14 || -11 - (j - j / 2) < -14) break;
                }
                if (run(kind, a, b, c, &r)) return 1;
                if (r == c + ulp) seen_pos = j;
                a[1] = -a[1];
                if (run(kind, a, b, c + ulp, &r)) return 1;
                if (r == c + ulp) seen_neg = j;
            }
            printf("    \"tie_breaker_seen_down_to_2^-j_ulp\": {\"positive\": %d, \"negative\": %d},\n", seen_pos, seen_neg);
        }
        // K - 1 products of 2^-j ulp each next to p0 = 0.5 ulp - (K - 1) 2^-j ulp ... simpler: many tiny products, do they add up?
        // p0 = 0.25 ulp, 15 x p = 1/32 ulp (sum 0.25 + 0.469 = 0.719 ulp > tie) -> c + ulp only if the tiny ones are summed
        if (K > 2) {
            std::vector<float> a(K, 0.0f), b(K, 0.0f);
            a[0] = 0.25f; b[0] = ulp;
            for (int k = 1; k < K; ++k) { a[k] = 0.03125f; b[k] = ulp; }
            if (kind == 2) { a[0] = 0.25f * ldexpf(1.0f, -10); b[0] = ldexpf(1.0f, -13); for (int k = 1; k < K; ++k) { a[k] = ldexpf(1.0f, -14); b[k] = ldexpf(1.0f, -14); } }
            if (run(kind, a, b, c, &r)) return 1;
            printf("    \"0.25ulp_plus_15_x_1_32ulp\": \"%a\", \"tiny_products_of_one_instruction_add_up\": %s,\n", r, r == c + ulp ? "true" : "false");
        }
        // empirical: 256-dim dots of positive values, chained exactly like the kernels do
        {
            std::mt19937 rng(1234 + kind);
            std::uniform_real_distribution<float> uni(0.5f, 1.0f);
            double worst = 0.0, sum_signed = 0.0;
            const int trials = 400;
            for (int t = 0; t < trials; ++t) {
                std::vector<float> a(256), b(256);
                for (int k = 0; k < 256; ++k) {
                    float x = uni(rng) / 16.0f, y = uni(rng) / 16.0f;
                    if (kind == 1) { uint32_t u; memcpy(&u, &x, 4); u &= 0xFFFF0000u; memcpy(&x, &u, 4); memcpy(&u, &y, 4); u &= 0xFFFF0000u; memcpy(&y, &u, 4); }
                    if (kind == 2) { x = (float)(_Float16)x; y = (float)(_Float16)y; }
                    a[k] = x; b[k] = y;
                }
                if (run(kind, a, b, 0.0f, &r)) return 1;
                double exact = 0.0;
                for (int k = 0; k < 256; ++k) exact += (double)a[k] * (double)b[k];
                const double rel = ((double)r - exact) / exact / ldexp(1.0, -24);
                worst = std::max(worst, std::fabs(rel));
                sum_signed += rel;
            }
            const int n_adds = kind == 0 ? 128 : 16;
            printf("    \"dot256_positive\": {\"chained_instructions\": %d, \"max_abs_error_in_units_of_2^-24_relative\": %.3f, "
                   "\"mean_signed_error_same_units\": %.3f, \"worst_case_if_every_instruction_rounds_to_nearest\": %d, \"trials\": %d},\n",
                   n_adds, worst, sum_signed / trials, n_adds, trials);
        }
        const char *mode = (up_075 && !up_025 && down_075 && !down_025) ? (ties_even ? "round-to-nearest-even" : "round-to-nearest (ties not to even)")
                           : (!up_075 && !up_025 && !down_075 && !down_025) ? "toward zero (truncation)"
                           : (!up_075 && !up_025 && down_075 && down_025) ? "toward -inf"
                           : (up_075 && up_025 && !down_075 && !down_025) ? "toward +inf" : "mixed";
        printf("    \"accumulate_rounding\": \"%s\"\n  }%s\n", mode, kind < 2 ? "," : "");
    }
    printf("}\n");
    return 0;
}
