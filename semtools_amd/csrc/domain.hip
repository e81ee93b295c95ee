// domain.hip -- the numeric DOMAIN of the library: which rows and queries the search kernels answer for, and the check that keeps
// everything else out of a corpus.
//
// The reference scores every (query, row) pair with simsimd's cosine and sorts with partial_cmp().unwrap_or(Equal)
// (src/search/mod.rs:86-89, 107-111).  For finite vectors of ordinary magnitude every simsimd backend agrees within 1e-5 and this
// library returns the f64 value of the "accurate" form bit for bit.  Outside that there is nothing definite to match:
//   * a NaN or Inf component makes ab / b2 non-finite and cos_finish's `unclipped > 0 ? unclipped : 0` turns the NaN into distance
//     0.0 -- the BEST score -- in every backend (oracle: tests/test_oracle.py::test_non_finite_rows_score_zero);
//   * components beyond ~1.8e19 overflow the f32 accumulators of the serial / SIMD backends (b2 = +Inf -> rsqrt 0 -> distance 1.0,
//     or NaN -> 0.0 when ab overflows too) while the f64 form still returns the true cosine; components below ~1e-19 underflow
//     them (b2 = 0 -> rsqrt +Inf -> 1 - Inf, clipped to distance 0.0).  Which rows those are depends on the backend's accumulation order
//     (16 partial sums overflow later than one), so the reference itself has no single answer there.
// The f32 / fp16 nominating kernels (K2, K3, K4, the operand image, the IVF index) and the certificate's error bounds
// (common.h F32_ERR_*) assume neither case.  So the boundary REFUSES such vectors instead of answering for them:
//
//   a row or query is IN DOMAIN iff every component is finite and its largest magnitude is 0 or lies in [2^-40, 2^40]
//
// (then sum x^2 <= 2^88 and >= 2^-80: no f32 overflow, no denormal sum, and scaling by a power of two changes no answer --
// tests/test_gpu_domain.py walks every kernel family at both ends).  Rows are checked where they ENTER a corpus
// (smt_corpus_append_host, smt_corpus_write_rows, smt_corpus_from_device, smt_embed(append_to), the file loaders and their sharded
// forms): SMT_E_INVALID, the corpus unchanged, the message names the first offending row.  Host-form queries are checked on the host;
// the device entry points report SMT_STATUS_INVALID_QUERY per query (final_select_kernel, scan_kernels.hip).
// model2vec's pool step with `normalize` emits unit or zero rows, so nothing the reference's own path produces is ever refused --
// only a corrupt table or hand-made vectors are.
#include "common.h"
#include "device_utils.h"

namespace smt {

// One wave per row: lane l holds components [4l, 4l + 4).  m = max |x| as an integer (IEEE bit patterns of non-negative floats are
// ordered; NaN and Inf sort above every finite value and fail the upper bound).  Bad rows are counted and the smallest index kept.
__global__ void __launch_bounds__(256) rows_domain_kernel(const float *rows, uint64_t n_rows, unsigned long long *out /* [count, first] */)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t r0 = wave * 4; r0 < n_rows; r0 += n_waves * 4) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t r = r0 + u < n_rows ? r0 + u : n_rows - 1;
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rows + r * 256) + lane);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint32_t m = max(max(v[u].x & 0x7fffffffu, v[u].y & 0x7fffffffu), max(v[u].z & 0x7fffffffu, v[u].w & 0x7fffffffu));
            m = max(m, dpp_u<DPP_XOR1>(m));
            m = max(m, dpp_u<DPP_XOR2>(m));
            m = max(m, dpp_u<DPP_HALF_MIRROR>(m));
            m = max(m, dpp_u<DPP_MIRROR>(m));
            const uint32_t m0 = (uint32_t)__builtin_amdgcn_readlane((int)m, 0), m1 = (uint32_t)__builtin_amdgcn_readlane((int)m, 16);
            const uint32_t m2 = (uint32_t)__builtin_amdgcn_readlane((int)m, 32), m3 = (uint32_t)__builtin_amdgcn_readlane((int)m, 48);
            const uint32_t mm = max(max(m0, m1), max(m2, m3));
            if (!magnitude_in_domain(mm) && lane == 0 && r0 + u < n_rows) {
                atomicAdd(out, 1ull);
                atomicMin(out + 1, (unsigned long long)(r0 + u));
            }
        }
    }
}

// n_rows rows at d_rows (device) checked on the context's stream; synchronises.  *first_bad = UINT64_MAX when all are in domain.
int check_rows_domain(smt_ctx *ctx, const float *d_rows, uint64_t n_rows, uint64_t *n_bad, uint64_t *first_bad)
{
    *n_bad = 0;
    *first_bad = ~0ull;
    if (n_rows == 0) return SMT_OK;
    unsigned long long init[2] = {0ull, ~0ull}, got[2] = {0ull, ~0ull};
    unsigned long long *d = ctx->d_status + 2;
    SMT_HIP_CHECK(hipMemcpyAsync(d, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    const uint64_t want = (n_rows + 15) / 16;   // 4 waves x 4 rows per block and step
    const int blocks = (int)std::min<uint64_t>(want, (uint64_t)(ctx->num_cus > 0 ? ctx->num_cus : 256) * 8);
    hipLaunchKernelGGL(rows_domain_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_rows, n_rows, d);
    SMT_HIP_CHECK(hipGetLastError());
    SMT_HIP_CHECK(hipMemcpyAsync(got, d, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *n_bad = got[0];
    *first_bad = got[1];
    return SMT_OK;
}

// ... and the error every ingest path returns for it: `what` names the entry point, `base` is added to the reported row
int require_rows_domain(smt_ctx *ctx, const float *d_rows, uint64_t n_rows, const char *what, uint64_t base)
{
    uint64_t n_bad = 0, first = 0;
    const int rc = check_rows_domain(ctx, d_rows, n_rows, &n_bad, &first);
    if (rc) return rc;
    if (n_bad == 0) return SMT_OK;
    set_error("%s: %llu of %llu rows are outside the library's domain (first: row %llu) -- every component must be finite and the "
              "largest magnitude of a row 0 or within [2^-40, 2^40] (include/semtools_hip.h, \"Domain\")",
              what, (unsigned long long)n_bad, (unsigned long long)n_rows, (unsigned long long)(base + first));
    return SMT_E_INVALID;
}

// The approximate index (ivfpq_*.hip) is built for what model2vec emits: UNIT rows (and zero rows, which every list scores alike).
// Its coarse quantiser, its residual codes and the ADC score q.c + sum LUT all work on the rows as they are, not on their
// directions, so rows of other lengths would be ranked by length as much as by angle (tests/test_gpu_domain.py: a power-of-two
// scaled corpus finds 23 of 200 true neighbours).  smt_ivfpq_build / _append therefore check | |x|^2 - 1 | <= 1e-3 (or
// x = 0) for every row the index is to cover and refuse the rest with SMT_E_UNSUPPORTED; the exact searches have no such condition.
__global__ void __launch_bounds__(256) rows_unit_kernel(const float *rows, uint64_t n_rows, unsigned long long *out /* [count, first] */)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t r0 = wave * 4; r0 < n_rows; r0 += n_waves * 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t r = r0 + u < n_rows ? r0 + u : n_rows - 1;
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(rows + r * 256) + lane);
        }
        float pb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pb[u] = v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
        const float b2 = wave_sum4(pb[0], pb[1], pb[2], pb[3], lane);   // lane l: |row l % 4|^2
        const bool bad = !(b2 == 0.0f || fabsf(b2 - 1.0f) <= 1.0e-3f) && r0 + (uint64_t)(lane & 3) < n_rows;
        if (bad && lane < 4) {
            atomicAdd(out, 1ull);
            atomicMin(out + 1, (unsigned long long)(r0 + (uint64_t)lane));
        }
    }
}

int require_unit_rows(smt_ctx *ctx, const float *d_rows, uint64_t n_rows, const char *what, uint64_t base)
{
    if (n_rows == 0) return SMT_OK;
    unsigned long long init[2] = {0ull, ~0ull}, got[2] = {0ull, ~0ull};
    unsigned long long *d = ctx->d_status + 2;
    SMT_HIP_CHECK(hipMemcpyAsync(d, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    const uint64_t want = (n_rows + 15) / 16;
    const int blocks = (int)std::min<uint64_t>(want, (uint64_t)(ctx->num_cus > 0 ? ctx->num_cus : 256) * 8);
    hipLaunchKernelGGL(rows_unit_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_rows, n_rows, d);
    SMT_HIP_CHECK(hipGetLastError());
    SMT_HIP_CHECK(hipMemcpyAsync(got, d, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (got[0] == 0) return SMT_OK;
    set_error("%s: %llu of %llu rows are not unit-length (first: row %llu) -- the approximate index covers unit and zero rows only "
              "(model2vec's normalised output); the exact searches take any row in the library's domain", what, got[0],
              (unsigned long long)n_rows, (unsigned long long)(base + got[1]));
    return SMT_E_UNSUPPORTED;
}

// host vectors (queries; rows that would overwrite live ones): index of the first one outside the domain, or -1
int64_t first_outside_domain_host(const float *v, uint64_t n, uint32_t dim)
{
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t m = 0;
        const float *x = v + i * dim;
        for (uint32_t d = 0; d < dim; ++d) {
            uint32_t b;
            memcpy(&b, x + d, 4);
            b &= 0x7fffffffu;
            m = b > m ? b : m;
        }
        if (!magnitude_in_domain(m)) return (int64_t)i;
    }
    return -1;
}

int require_queries_domain_host(const float *queries, uint32_t nq, const char *what)
{
    const int64_t bad = first_outside_domain_host(queries, nq, SMT_DIM);
    if (bad < 0) return SMT_OK;
    set_error("%s: query %lld is outside the library's domain -- every component must be finite and the largest magnitude 0 or "
              "within [2^-40, 2^40] (include/semtools_hip.h, \"Domain\")", what, (long long)bad);
    return SMT_E_INVALID;
}

}  // namespace smt
