/*
 * cpu_fast.c -- "fair CPU" scan used ONLY as bench.py's cpu_baseline and by
 * tests as a cross-check.  TEST INFRASTRUCTURE, NOT a restatement of the
 * reference: the reference loop (src/search/mod.rs:84-119) is single-threaded,
 * materialises every row's result and sorts them all; this variant threads the
 * row loop (OpenMP), lets the compiler vectorise the 256-d dot product
 * (-O3 -mavx2 -mfma: summation order differs from the serial oracle at the
 * 1e-7 level) and keeps a bounded per-thread list.  Same selection rule:
 * (distance asc, row asc) == the reference's stable sort order.
 */
#include "semtools_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double d; uint64_t row; } cand;

static inline int cand_less(const cand *a, const cand *b)
{
    return a->d < b->d || (a->d == b->d && a->row < b->row);
}

static inline void list_insert(cand *list, uint64_t *n, uint64_t k, cand c)
{
    if (*n == k && !cand_less(&c, &list[k - 1])) return;
    uint64_t pos = *n < k ? *n : k - 1;
    while (pos > 0 && cand_less(&c, &list[pos - 1])) { list[pos] = list[pos - 1]; --pos; }
    list[pos] = c;
    if (*n < k) ++*n;
}

uint64_t orc_scan_topk_threads(const float *emb, uint64_t N, uint32_t D,
                               const float *query, uint64_t top_k, int n_threads,
                               uint64_t *out_rows, double *out_dist)
{
    if (top_k == 0 || N == 0) return 0;
    if (n_threads < 1) n_threads = 1;
    float a2 = 0.0f;
    for (uint32_t i = 0; i < D; ++i) a2 += query[i] * query[i];
    cand *lists = (cand *)malloc((size_t)n_threads * top_k * sizeof(cand));
    uint64_t *counts = (uint64_t *)calloc((size_t)n_threads, sizeof(uint64_t));
#pragma omp parallel num_threads(n_threads)
    {
        int t = 0, T = 1;
#ifdef _OPENMP
        t = omp_get_thread_num();
        T = omp_get_num_threads();
#endif
        uint64_t r0 = N * (uint64_t)t / (uint64_t)T, r1 = N * (uint64_t)(t + 1) / (uint64_t)T;
        cand *list = lists + (size_t)t * top_k;
        uint64_t n = 0;
        for (uint64_t r = r0; r < r1; ++r) {
            const float *e = emb + r * (uint64_t)D;
            float ab = 0.0f, b2 = 0.0f;
#pragma omp simd reduction(+ : ab, b2)
            for (uint32_t i = 0; i < D; ++i) { ab += query[i] * e[i]; b2 += e[i] * e[i]; }
            double d;
            if (a2 == 0 && b2 == 0) d = 0.0;
            else if (ab == 0) d = 1.0;
            else {
                d = 1.0 - (double)ab * (1.0 / sqrt((double)a2)) * (1.0 / sqrt((double)b2));
                if (!(d > 0)) d = 0.0;
            }
            cand c = { d, r };
            list_insert(list, &n, top_k, c);
        }
        counts[t] = n;
    }
    cand *fin = (cand *)malloc(top_k * sizeof(cand));
    uint64_t nf = 0;
    for (int t = 0; t < n_threads; ++t)
        for (uint64_t j = 0; j < counts[t]; ++j)
            list_insert(fin, &nf, top_k, lists[(size_t)t * top_k + j]);
    for (uint64_t j = 0; j < nf; ++j) { out_rows[j] = fin[j].row; out_dist[j] = fin[j].d; }
    free(fin); free(lists); free(counts);
    return nf;
}

/* ---------------------------------------------------------------------------
 * orc_search_documents_simd -- the reference's CONTROL FLOW (src/search/mod.rs:84-119: one cosine call per row,
 * a result record for EVERY row under the threshold -- default 100.0, i.e. all of them --, a stable sort of all
 * records, take(top_k)) with the per-row cosine the reference actually runs on a modern x86 host: simsimd 6.5.1
 * dispatches at run time to its AVX-512 (skylake) or AVX2+FMA (haswell) f32 kernel, which accumulates ab, a2 and
 * b2 in f32 lanes in ONE pass over the two vectors.  Single thread, like the reference.  This is bench.py's
 * honest single-core CPU baseline ("port-simd"); the scalar restatement in semtools_oracle.c is 5-10x slower per
 * row than what the reference executes.  What is NOT modelled: the String clones of the file name and of the
 * context lines that the reference attaches to every record (mod.rs:93-100) -- the real thing is slower still.
 * TEST INFRASTRUCTURE / reported baseline only.
 */
#include <immintrin.h>

__attribute__((target("avx512f"))) static void cos3_avx512(const float *a, const float *b, uint32_t n, float *ab, float *a2, float *b2)
{
    __m512 vab = _mm512_setzero_ps(), va2 = _mm512_setzero_ps(), vb2 = _mm512_setzero_ps();
    uint32_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m512 x = _mm512_loadu_ps(a + i), y = _mm512_loadu_ps(b + i);
        vab = _mm512_fmadd_ps(x, y, vab);
        va2 = _mm512_fmadd_ps(x, x, va2);
        vb2 = _mm512_fmadd_ps(y, y, vb2);
    }
    float sab = _mm512_reduce_add_ps(vab), sa2 = _mm512_reduce_add_ps(va2), sb2 = _mm512_reduce_add_ps(vb2);
    for (; i < n; ++i) { sab += a[i] * b[i]; sa2 += a[i] * a[i]; sb2 += b[i] * b[i]; }
    *ab = sab; *a2 = sa2; *b2 = sb2;
}

__attribute__((target("avx2,fma"))) static void cos3_avx2(const float *a, const float *b, uint32_t n, float *ab, float *a2, float *b2)
{
    __m256 vab = _mm256_setzero_ps(), va2 = _mm256_setzero_ps(), vb2 = _mm256_setzero_ps();
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256 x = _mm256_loadu_ps(a + i), y = _mm256_loadu_ps(b + i);
        vab = _mm256_fmadd_ps(x, y, vab);
        va2 = _mm256_fmadd_ps(x, x, va2);
        vb2 = _mm256_fmadd_ps(y, y, vb2);
    }
    float t[8], sab = 0.f, sa2 = 0.f, sb2 = 0.f;
    _mm256_storeu_ps(t, vab); for (int j = 0; j < 8; ++j) sab += t[j];
    _mm256_storeu_ps(t, va2); for (int j = 0; j < 8; ++j) sa2 += t[j];
    _mm256_storeu_ps(t, vb2); for (int j = 0; j < 8; ++j) sb2 += t[j];
    for (; i < n; ++i) { sab += a[i] * b[i]; sa2 += a[i] * a[i]; sb2 += b[i] * b[i]; }
    *ab = sab; *a2 = sa2; *b2 = sb2;
}

typedef void (*cos3_fn)(const float *, const float *, uint32_t, float *, float *, float *);

const char *orc_simd_backend(void)
{
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx512f") ? "avx512f" : "avx2+fma";
}

static orc_result *merge_sort_results(orc_result *v, orc_result *tmp, uint64_t n)
{
    /* bottom-up stable merge sort by distance, ping-pong between the two buffers (Rust's sort_by is a stable
     * merge sort as well; partial_cmp, NaN => Equal).  Returns the buffer that holds the sorted records. */
    orc_result *src = v, *dst = tmp;
    for (uint64_t w = 1; w < n; w *= 2) {
        for (uint64_t lo = 0; lo < n; lo += 2 * w) {
            uint64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            uint64_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) dst[o++] = (src[j].distance < src[i].distance) ? src[j++] : src[i++];
            while (i < mid) dst[o++] = src[i++];
            while (j < hi) dst[o++] = src[j++];
        }
        orc_result *t = src; src = dst; dst = t;
    }
    return src;
}

uint64_t orc_search_documents_simd(const float *emb, uint64_t N, uint32_t D, const float *query, uint64_t n_lines,
                                   uint64_t top_k, int has_max_distance, double max_distance, orc_result *out, uint64_t cap)
{
    __builtin_cpu_init();
    const cos3_fn cos3 = __builtin_cpu_supports("avx512f") ? cos3_avx512 : cos3_avx2;
    const double thr = has_max_distance ? max_distance : 100.0;        /* mod.rs:88 unwrap_or(100.0) */
    orc_result *all = (orc_result *)malloc((size_t)(N ? N : 1) * sizeof(orc_result));
    uint64_t n = 0;
    for (uint64_t r = 0; r < N; ++r) {                                  /* mod.rs:84-85 (one document) */
        float ab, a2, b2;
        cos3(query, emb + r * (uint64_t)D, D, &ab, &a2, &b2);           /* mod.rs:86: all three sums per call */
        double d;
        if (a2 == 0 && b2 == 0) d = 0.0;
        else if (ab == 0) d = 1.0;
        else {
            d = 1.0 - (double)ab * (1.0 / sqrt((double)a2)) * (1.0 / sqrt((double)b2));
            if (!(d > 0)) d = 0.0;
        }
        if (d < thr) {                                                  /* mod.rs:89 */
            orc_result x;
            x.doc = 0;
            x.match_line = r;
            x.start = r >= n_lines ? r - n_lines : 0;                   /* mod.rs:90 */
            x.end = r + n_lines + 1 < N ? r + n_lines + 1 : N;          /* mod.rs:91 */
            x.distance = d;
            all[n++] = x;                                               /* mod.rs:93-100 (minus the String clones) */
        }
    }
    orc_result *tmp = (orc_result *)malloc((size_t)(n ? n : 1) * sizeof(orc_result));
    const orc_result *sorted = merge_sort_results(all, tmp, n);          /* mod.rs:107-111 */
    const uint64_t keep = has_max_distance ? n : (n < top_k ? n : top_k); /* mod.rs:115-119 */
    for (uint64_t i = 0; i < keep && i < cap; ++i) out[i] = sorted[i];
    free(tmp);
    free(all);
    return keep;
}
