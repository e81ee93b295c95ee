/*
 * semtools_oracle.c -- CPU restatement of the semtools search hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see semtools_oracle.h for who may use it and for
 * the "parity unpinned" statement).  Build: oracle/Makefile.  The strict
 * functions are compiled -O2 -ffp-contract=off so every f32 operation is one
 * IEEE-754 rounding in source order (no FMA, no reassociation) -- that is what
 * a baseline x86-64 build of the upstream crates executes.
 *
 * Citations "file:line" are relative to /root/reference (semtools v3.0.0).
 * [UPSTREAM-RECALL] marks algorithm text restated from the published crates
 * (sources absent from this container).
 */
#include "semtools_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ===================================================================== A3/A4
 * model2vec-rs 0.1.3, StaticModel::pool_ids [UPSTREAM-RECALL]:
 *     let mut sum = vec![0.0; dim];
 *     for &id in &ids { for (i,&v) in row(id).iter().enumerate() { sum[i] += v; } }
 *     let cnt = ids.len().max(1) as f32;
 *     sum.iter_mut().for_each(|v| *v /= cnt);
 *     if normalize { let norm = sum.iter().map(|&v| v*v).sum::<f32>().sqrt().max(1e-12);
 *                    sum.iter_mut().for_each(|v| *v /= norm); }
 * encode_with_args truncates the (unk-filtered) id list to max_length first.
 * Call sites: src/search/mod.rs:69 (max_length 2048), src/cmds/search.rs:136
 * (encode_single -> 512).  Ids >= V cannot occur upstream (tokenizer vocab ==
 * table rows); here they are skipped so a bad test input cannot read OOB.
 */
void orc_pool_ids(const float *table, uint64_t V, uint32_t D, int normalize,
                  const uint32_t *ids, uint64_t n_ids, uint32_t max_tokens,
                  float *out)
{
    uint64_t n = n_ids;
    if (max_tokens != 0 && n > (uint64_t)max_tokens) n = max_tokens;
    for (uint32_t d = 0; d < D; ++d) out[d] = 0.0f;
    uint64_t used = 0;
    for (uint64_t t = 0; t < n; ++t) {
        uint64_t id = ids[t];
        if (id >= V) continue;
        const float *row = table + id * (uint64_t)D;
        for (uint32_t d = 0; d < D; ++d) out[d] = out[d] + row[d];
        ++used;
    }
    (void)used;
    float cnt = (float)(n > 0 ? n : 1);
    for (uint32_t d = 0; d < D; ++d) out[d] = out[d] / cnt;
    if (normalize) {
        float ss = 0.0f;
        for (uint32_t d = 0; d < D; ++d) ss = ss + out[d] * out[d];
        float norm = sqrtf(ss);
        if (!(norm > 1e-12f)) norm = 1e-12f; /* f32::max(1e-12) */
        for (uint32_t d = 0; d < D; ++d) out[d] = out[d] / norm;
    }
}

void orc_embed_lines(const float *table, uint64_t V, uint32_t D, int normalize,
                     const uint32_t *ids, const uint64_t *offsets,
                     uint64_t n_lines, uint32_t max_tokens, float *out)
{
    for (uint64_t i = 0; i < n_lines; ++i)
        orc_pool_ids(table, V, D, normalize, ids + offsets[i],
                     offsets[i + 1] - offsets[i], max_tokens, out + i * (uint64_t)D);
}

/* ======================================================================= A5
 * simsimd 6.5.1 include/simsimd/spatial.h, SIMSIMD_MAKE_COS [UPSTREAM-RECALL]:
 *     ab += ai*bi, a2 += ai*ai, b2 += bi*bi   (accumulator type per backend)
 *     if (a2 == 0 && b2 == 0) result = 0;
 *     else if (ab == 0)       result = 1;
 *     else { unclipped = 1 - ab * RSQRT(a2) * RSQRT(b2);
 *            result = unclipped > 0 ? unclipped : 0; }
 * result type simsimd_distance_t = f64; the Rust binding returns Option<f64>
 * (None only on length mismatch).  Call site: src/search/mod.rs:86.
 * The SIMD backends the dispatcher would pick on a given CPU accumulate in a
 * different order and use rsqrt+Newton, which is why the contract on distances
 * is 1e-5 and not bit-exactness.
 */
static double cos_finish(double ab, double a2, double b2)
{
    if (a2 == 0 && b2 == 0) return 0.0;
    if (ab == 0) return 1.0;
    double unclipped = 1.0 - ab * (1.0 / sqrt(a2)) * (1.0 / sqrt(b2));
    return unclipped > 0 ? unclipped : 0.0;
}

double orc_cosine_f32_serial(const float *a, const float *b, uint32_t n)
{
    float ab = 0.0f, a2 = 0.0f, b2 = 0.0f;
    for (uint32_t i = 0; i < n; ++i) {
        float ai = a[i], bi = b[i];
        ab = ab + ai * bi;
        a2 = a2 + ai * ai;
        b2 = b2 + bi * bi;
    }
    return cos_finish((double)ab, (double)a2, (double)b2);
}

double orc_cosine_f32_accurate(const float *a, const float *b, uint32_t n)
{
    double ab = 0.0, a2 = 0.0, b2 = 0.0;
    for (uint32_t i = 0; i < n; ++i) {
        double ai = a[i], bi = b[i];
        ab = ab + ai * bi;
        a2 = a2 + ai * ai;
        b2 = b2 + bi * bi;
    }
    return cos_finish(ab, a2, b2);
}

/* ======================================================================= A6
 * search_documents, src/search/mod.rs:77-120:
 *   for doc, for (idx, line_embedding): distance = f32::cosine(q, e)   :84-86
 *   threshold = max_distance.unwrap_or(100.0); keep if distance < threshold :88-89
 *   bottom = idx.saturating_sub(n_lines); top = min(len, idx + n_lines + 1) :90-91
 *   push result                                                         :93-100
 *   stable sort by partial_cmp (NaN => Equal)                           :107-111
 *   max_distance.is_some() ? all : take(top_k)                          :115-119
 */
static int res_less(const orc_result *a, const orc_result *b)
{
    /* partial_cmp(..).unwrap_or(Equal) == Less */
    return a->distance < b->distance;
}

static void res_merge_sort(orc_result *v, orc_result *tmp, uint64_t n)
{
    if (n < 2) return;
    uint64_t h = n / 2;
    res_merge_sort(v, tmp, h);
    res_merge_sort(v + h, tmp, n - h);
    uint64_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* stable: take from the right run only if strictly less */
        if (res_less(&v[j], &v[i])) tmp[k++] = v[j++];
        else tmp[k++] = v[i++];
    }
    while (i < h) tmp[k++] = v[i++];
    while (j < n) tmp[k++] = v[j++];
    memcpy(v, tmp, n * sizeof(orc_result));
}

uint64_t orc_search_documents(const float *emb, const uint64_t *doc_line_counts,
                              uint64_t n_docs, uint32_t D, const float *query,
                              uint64_t n_lines, uint64_t top_k, int has_max_distance,
                              double max_distance, int accurate, orc_result *out,
                              uint64_t cap)
{
    uint64_t total = 0;
    for (uint64_t d = 0; d < n_docs; ++d) total += doc_line_counts[d];
    orc_result *res = (orc_result *)malloc((total ? total : 1) * sizeof(orc_result));
    orc_result *tmp = (orc_result *)malloc((total ? total : 1) * sizeof(orc_result));
    if (!res || !tmp) { free(res); free(tmp); return 0; }
    const double threshold = has_max_distance ? max_distance : 100.0;
    uint64_t n = 0, row = 0;
    for (uint64_t d = 0; d < n_docs; ++d) {
        uint64_t len = doc_line_counts[d];
        for (uint64_t idx = 0; idx < len; ++idx, ++row) {
            const float *e = emb + row * (uint64_t)D;
            double dist = accurate ? orc_cosine_f32_accurate(query, e, D)
                                   : orc_cosine_f32_serial(query, e, D);
            if (dist < threshold) {
                uint64_t bottom = idx > n_lines ? idx - n_lines : 0;
                uint64_t top = idx + n_lines + 1;
                if (top > len) top = len;
                res[n].doc = d;
                res[n].match_line = idx;
                res[n].start = bottom;
                res[n].end = top;
                res[n].distance = dist;
                ++n;
            }
        }
    }
    res_merge_sort(res, tmp, n);
    uint64_t ret = has_max_distance ? n : (n < top_k ? n : top_k);
    uint64_t w = ret < cap ? ret : cap;
    if (out && w) memcpy(out, res, w * sizeof(orc_result));
    free(res);
    free(tmp);
    return ret;
}

/* ====================================================================== A10
 * Store::search_line_embeddings, src/workspace/store.rs:481-546:
 *   empty subset or top_k == 0 -> []                                   :489-491
 *   per chunk of 1000 paths: Nearest(q), filter path in chunk,
 *     score_threshold = 1 - max_distance, limit = 2*top_k              :495-523
 *   distance = 1 - score (f32)                                         :531
 *   stable sort asc by distance, truncate(top_k)                       :538-543
 * qdrant-edge internals [UPSTREAM-RECALL]: Distance::Cosine collections
 * L2-normalise vectors on insert and the query on search (cosine_preprocess:
 * len2 = sum x*x; unchanged if len2 < f32::EPSILON or |len2-1| <= 1e-6, else
 * x / sqrt(len2)); score = dot product in f32; the default ("Plain") index is
 * an exact filtered scan; score_threshold keeps score > threshold for
 * similarity metrics; top-`limit` by score descending.  Tie order between
 * equal scores is unspecified upstream; here earlier storage row wins.
 */
static void qdrant_cosine_preprocess(const float *x, uint32_t D, float *y)
{
    float len2 = 0.0f;
    for (uint32_t i = 0; i < D; ++i) len2 = len2 + x[i] * x[i];
    if (len2 < 1.1920929e-07f || fabsf(len2 - 1.0f) <= 1.0e-6f) {
        memcpy(y, x, D * sizeof(float));
        return;
    }
    float len = sqrtf(len2);
    for (uint32_t i = 0; i < D; ++i) y[i] = x[i] / len;
}

typedef struct { float score; uint64_t row; } scored_row;

uint64_t orc_search_line_embeddings(const float *emb, const uint32_t *row_path,
                                    const int32_t *row_line, uint64_t N, uint32_t D,
                                    const float *query, const uint32_t *subset,
                                    uint64_t n_subset, uint64_t top_k,
                                    int has_max_distance, float max_distance,
                                    orc_ranked_line *out, uint64_t cap)
{
    if (n_subset == 0 || top_k == 0) return 0;
    float *q = (float *)malloc(D * sizeof(float));
    float *v = (float *)malloc(D * sizeof(float));
    uint32_t max_path = 0;
    for (uint64_t i = 0; i < N; ++i) if (row_path[i] > max_path) max_path = row_path[i];
    for (uint64_t i = 0; i < n_subset; ++i) if (subset[i] > max_path) max_path = subset[i];
    uint8_t *in_chunk = (uint8_t *)calloc((size_t)max_path + 1, 1);
    const uint64_t limit = top_k * 2;
    scored_row *best = (scored_row *)malloc((limit + 1) * sizeof(scored_row));
    uint64_t n_all = 0, cap_all = 0;
    orc_ranked_line *all = NULL;
    qdrant_cosine_preprocess(query, D, q);
    const float score_threshold = 1.0f - max_distance;

    for (uint64_t c0 = 0; c0 < n_subset; c0 += 1000) {
        uint64_t c1 = c0 + 1000 < n_subset ? c0 + 1000 : n_subset;
        memset(in_chunk, 0, (size_t)max_path + 1);
        for (uint64_t i = c0; i < c1; ++i) in_chunk[subset[i]] = 1;
        uint64_t nb = 0;
        for (uint64_t r = 0; r < N; ++r) {
            if (!in_chunk[row_path[r]]) continue;
            qdrant_cosine_preprocess(emb + r * (uint64_t)D, D, v);
            float s = 0.0f;
            for (uint32_t i = 0; i < D; ++i) s = s + q[i] * v[i];
            if (has_max_distance && !(s > score_threshold)) continue;
            /* keep the `limit` best by (score desc, row asc) */
            uint64_t pos = nb;
            while (pos > 0 && best[pos - 1].score < s) --pos;
            if (pos >= limit) continue;
            uint64_t last = nb < limit ? nb : limit - 1;
            for (uint64_t j = last; j > pos; --j) best[j] = best[j - 1];
            best[pos].score = s;
            best[pos].row = r;
            if (nb < limit) ++nb;
        }
        if (n_all + nb > cap_all) {
            cap_all = (n_all + nb) * 2 + 16;
            all = (orc_ranked_line *)realloc(all, cap_all * sizeof(orc_ranked_line));
        }
        for (uint64_t j = 0; j < nb; ++j) {
            all[n_all].path_id = row_path[best[j].row];
            all[n_all].line_number = row_line[best[j].row];
            all[n_all].distance = 1.0f - best[j].score;
            all[n_all].row = best[j].row;
            ++n_all;
        }
    }
    /* stable insertion sort asc by distance (n_all <= 2*top_k*chunks) */
    for (uint64_t i = 1; i < n_all; ++i) {
        orc_ranked_line x = all[i];
        uint64_t j = i;
        while (j > 0 && x.distance < all[j - 1].distance) { all[j] = all[j - 1]; --j; }
        all[j] = x;
    }
    uint64_t ret = n_all < top_k ? n_all : top_k;
    uint64_t w = ret < cap ? ret : cap;
    if (out && w) memcpy(out, all, w * sizeof(orc_ranked_line));
    free(all); free(best); free(in_chunk); free(q); free(v);
    return ret;
}

/* ======================================================================== A9
 * fnv1a_hash, src/workspace/store.rs:651-661; LineEmbedding::id :82-89
 * (path bytes || line_number.to_le_bytes()); DocMeta::id :75-80.
 */
uint64_t orc_fnv1a_hash(const uint8_t *bytes, uint64_t n)
{
    uint64_t hash = 0xcbf29ce484222325ULL;
    for (uint64_t i = 0; i < n; ++i) {
        hash ^= (uint64_t)bytes[i];
        hash *= 0x100000001b3ULL;
    }
    return hash;
}

uint64_t orc_doc_meta_id(const char *path)
{
    return orc_fnv1a_hash((const uint8_t *)path, strlen(path));
}

uint64_t orc_line_embedding_id(const char *path, int32_t line_number)
{
    size_t n = strlen(path);
    uint8_t *buf = (uint8_t *)malloc(n + 4);
    memcpy(buf, path, n);
    uint32_t u = (uint32_t)line_number;
    buf[n + 0] = (uint8_t)(u & 0xff);
    buf[n + 1] = (uint8_t)((u >> 8) & 0xff);
    buf[n + 2] = (uint8_t)((u >> 16) & 0xff);
    buf[n + 3] = (uint8_t)((u >> 24) & 0xff);
    uint64_t h = orc_fnv1a_hash(buf, n + 4);
    free(buf);
    return h;
}
