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
