// ivfpq.h -- IVF-PQ index over the resident corpus (BASELINE config 5).
//
// NO reference counterpart: the reference's workspace store is an exact scan (its "IVF_PQ"/"HNSW"
// labels are cosmetic).  The contract is therefore recall against this library's own
// exact path, not parity with reference code.  Candidates found through the index are ALWAYS
// re-ranked with the exact f64 distance of the exact path (final_select_kernel), so every returned
// (row, distance) pair is a true pair -- only membership of the top-k is approximate.
//
// Build (all on the GPU):
//   coarse k-means (nlist centroids, D=256, L2): assignment = f32-MFMA C x centroids^T with the same
//     corpus-stationary tiling as K3 (row tile in registers, centroid tiles through LDS); each lane
//     keeps a running arg-max per row over the centroids it sees, one cross-lane reduce per row tile;
//     update = fixed-point (2^-32) integer atomics => order-independent, bit-reproducible centroids;
//   product quantiser: m=32 subspaces x 8 dims x 256 codes trained on residuals (x - centroid),
//     codebooks staged in LDS as [code][subspace][8] (conflict-free for lanes = subspaces);
//   inverted lists: rocPRIM radix sort of (list id, row), codes written in list order (32 B/row).
// Query:
//   probe: 0.5|c|^2 - q.c for all (query, centroid) pairs on the MFMA pipe, then one wave per query selects the
//     nprobe smallest of its 4096 scores held in registers (bisection on the orderable bit pattern);
//   LUT[s][code] = <q_s, codebook[s][code]> (inner product: rows are unit-norm, so the ADC score
//     q.c_list + sum_s LUT[s][code_s] approximates cos(q, x)); 32 KiB per query, LDS resident;
//   ADC scan: one block per (query, probed list) streams 32-B codes (nprobe/nlist * N * 32 B per query),
//     32 LDS lookups per row; each wave keeps its best candidates by threshold selection in registers,
//     (optionally prunes them with an int8 copy of the rows,) and re-scores them against the full-precision rows;
//   select: the K2 select stage merges the per-list candidate lists and rescoring is EXACT.

//
// Files: ivfpq_build.hip (coarse k-means, quantiser training, encoding, inverted lists; build and append), ivfpq_search.hip (probe,
// LUT / projection, ADC scan, the search entry points), ivfpq_io.hip (destroy, info, save, load); this header: what they share.
#pragma once
#include <algorithm>
#include <memory>
#include <vector>

#include "common.h"
#include "device_utils.h"
#include "mfma_tile.h"

namespace smt {

constexpr int PQ_M = 32;      // subspaces
constexpr int PQ_DSUB = 8;    // dims per subspace (256 / 32)
constexpr int PQ_K = 256;     // codes per subspace (8 bits)
constexpr double FIXED_SCALE = 4294967296.0;  // 2^32

__device__ __forceinline__ uint32_t f32_orderable(float f)
{
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);  // ascending u32 == ascending float
}

constexpr int PROBE_THREADS = 1024;
constexpr int PROBE_MAX_LISTS = 4096;     // the probe selection holds a query's scores in the registers of one wave (ivfpq_search.hip)

// per-list PCA codes (index kind 1; ivfpq_build.hip lpca_train_kernel)
constexpr int LP_DIMS = 32;            // directions kept per list == code bytes per row
constexpr int LP_TILE = 32;            // residual rows per tile
constexpr int LP_RSTRIDE = 257;        // LDS row stride of the residual tile (conflict-free column walks)
constexpr int LP_TRAIN_ROWS = 768;     // rows of a list the basis is fitted to (evenly spaced sample)

}  // namespace smt

struct smt_ivfpq {
    smt_corpus *corpus = nullptr;
    int device = -1;                // the corpus' GPU (recorded so that destroy never has to look at the corpus)
    uint64_t n_rows = 0;
    uint32_t nlist = 0;
    float *d_centroids = nullptr;   // [nlist][256]
    float *d_cnorm_half = nullptr;  // [nlist]
    float *d_codebooks = nullptr;   // [32][256][8]
    uint8_t *d_codes = nullptr;     // [N][32] list order
    uint32_t *d_ids = nullptr;      // [N]
    uint64_t *d_offsets = nullptr;  // [nlist+1]
    uint32_t kind = 0;              // 0: global residual codebooks (dsub 8); 1: per-list PCA basis + 8-bit scalar codes
    float *d_basis = nullptr;       // kind 1: [nlist][32][256]
    float *d_lscale = nullptr;      // kind 1: [nlist][32]
    uint64_t max_list = 0;          // longest inverted list (segments per probed list at query time)
    double build_ms[4] = {0, 0, 0, 0};  // coarse train, assign all, pq train, encode+lists
};

#define IVF_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            smt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SMT_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)

namespace smt {

struct IvfDevBuf {
    void *p = nullptr;
    ~IvfDevBuf() { if (p) (void)hipFree(p); }
    template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};
inline int ivf_dev_alloc(IvfDevBuf &b, size_t bytes)
{
    hipError_t e = hipMalloc(&b.p, bytes ? bytes : 16);
    if (e != hipSuccess) { set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return SMT_E_NOMEM; }
    return SMT_OK;
}

int ivf_compute_max_list(smt_ivfpq *ix);   // longest inverted list, from the offsets on the device (ivfpq_io.hip)

}  // namespace smt
