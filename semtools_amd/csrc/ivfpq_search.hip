// ivfpq_search.hip -- querying the IVF index: probe (query x centroid scores on the MFMA pipe, nprobe smallest per query), LUT or
// per-list projection, the ADC scan with its in-kernel re-score of the shortlist, and the search entry points.  ivfpq.h, DESIGN.md 4.6.
#include <atomic>
#include <chrono>

#include "ivfpq.h"

namespace smt {

// ------------------------------------------------------------------ query: probe

struct ProbeParams {
    const float *queries;     // [nq][256]
    const float *centroids;   // [nlist][256]
    const float *cnorm_half;
    uint32_t nlist;           // <= 4096
    uint32_t nprobe;
    uint32_t *probe_list;     // [nq][nprobe]
    float *probe_dot;         // [nq][nprobe]  q . c
};

// Coarse probe, two kernels (the first version was one block per query: 4096 wave-level dot products against
// centroids re-read from L2 by every query, then a 78-stage bitonic sort of all 4096 keys -- 0.45 ms per 1000
// queries, as much as the ADC scan itself):
//   ivf_score_kernel   S[q][c] = 0.5|c|^2 - q.c for all (query, centroid) pairs on the MFMA pipe; one block per
//                      centroid tile (32 centroids staged in LDS once), its waves sweep the query tiles;
//   ivf_probe_select_kernel   one WAVE per query holds its nlist scores in registers (64 per lane), finds the
//                      nprobe-th smallest by bisection on the orderable bit pattern, and emits the nprobe lists
//                      (order is irrelevant downstream; ties go to the smaller list id).
__global__ void __launch_bounds__(GEMM_THREADS, 2) ivf_score_kernel(const float *queries, uint32_t nq, const float *centroids,
                                                                    const float *cnorm_half, uint32_t nlist, float *scores)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4 *s_c = reinterpret_cast<f32x4 *>(smem_raw);  // [32][65] float4: this block's centroid tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const uint32_t ct = blockIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = threadIdx.x + u * GEMM_THREADS;
        s_c[(idx >> 6) * QT_STRIDE_F4 + (idx & 63)] =
            reinterpret_cast<const f32x4 *>(centroids + (size_t)(ct * QT_ROWS + (idx >> 6)) * 256)[idx & 63];
    }
    const uint32_t cid = ct * QT_ROWS + j;
    const float cn = cnorm_half[cid];
    __syncthreads();
    const uint32_t n_tiles = (nq + 31) / 32;
    for (uint32_t tile = wave; tile < n_tiles; tile += GEMM_WAVES) {
        const uint32_t qrow = tile * 32 + j;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(queries + (size_t)(qrow < nq ? qrow : 0) * 256) + h;
        f32x4 A[32];
#pragma unroll
        for (int m = 0; m < 32; ++m) A[m] = src[2 * m];
        const f32x16 acc = mfma_tile_32x32x256(A, s_c + j * QT_STRIDE_F4 + h);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t row = tile * 32 + acc_row(r, h);
            if (row < nq) scores[(size_t)row * nlist + cid] = cn - acc[r];
        }
    }
}

constexpr int SEL_SLOTS = PROBE_MAX_LISTS / 64;  // scores per lane

__global__ void __launch_bounds__(256) ivf_probe_select_kernel(ProbeParams p, const float *scores, uint32_t nq)
{
    const int lane = threadIdx.x & 63;
    const uint32_t qi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (qi >= nq) return;  // wave-uniform
    const float *sc = scores + (size_t)qi * p.nlist;
    uint32_t v[SEL_SLOTS];
#pragma unroll
    for (int u = 0; u < SEL_SLOTS; ++u) {
        const uint32_t c = (uint32_t)u * 64u + (uint32_t)lane;
        v[u] = c < p.nlist ? f32_orderable(sc[c]) : 0xFFFFFFFFu;
    }
    // smallest T with #(v <= T) >= nprobe
    uint32_t lo = 0u, hi = 0xFFFFFFFFu;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        uint32_t cnt = 0;
#pragma unroll
        for (int u = 0; u < SEL_SLOTS; ++u) cnt += v[u] <= mid ? 1u : 0u;
        cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_sum_u32(cnt));
        if (cnt >= p.nprobe) hi = mid; else lo = mid + 1u;
    }
    const uint32_t T = lo;
    uint32_t n_lt = 0;
#pragma unroll
    for (int u = 0; u < SEL_SLOTS; ++u) n_lt += v[u] < T ? 1u : 0u;
    n_lt = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_sum_u32(n_lt));
    uint32_t pos_lt = 0, pos_eq = n_lt;  // wave-uniform write cursors: "< T" first, then ties in list-id order
    uint32_t *out_l = p.probe_list + (size_t)qi * p.nprobe;
    float *out_d = p.probe_dot + (size_t)qi * p.nprobe;
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int u = 0; u < SEL_SLOTS; ++u) {
        const uint32_t c = (uint32_t)u * 64u + (uint32_t)lane;
        const bool lt = v[u] < T, eq = v[u] == T && c < p.nlist;
        const unsigned long long m_lt = __ballot(lt), m_eq = __ballot(eq);
        uint32_t slot = 0xFFFFFFFFu;
        if (lt) slot = pos_lt + (uint32_t)__popcll(m_lt & below);
        else if (eq) slot = pos_eq + (uint32_t)__popcll(m_eq & below);
        if (slot < p.nprobe) {
            out_l[slot] = c;
            out_d[slot] = p.cnorm_half[c] - sc[c];  // q . c
        }
        pos_lt += (uint32_t)__popcll(m_lt);
        pos_eq += (uint32_t)__popcll(m_eq);
    }
}

// LUT[q][code][s] = <q_s, codebook[s][code]>; grid (nq, 32), 256 threads.  CODE-major: sub-quantiser s owns LDS bank s of the scan
// kernel's copy (ivf_adc_kernel, KIND 0), whatever the codes are.
__global__ void ivf_lut_kernel(const float *queries, const float *codebooks, float *lut)
{
    // 8 codes x 32 sub-quantisers per block, s fastest: the stores are coalesced (with one sub-quantiser per block and code = thread
    // they lay 128 B apart and this kernel doubled the probe stage, 0.11 -> 0.21 ms per 1000 queries); the 32 B codebook reads scatter
    // over the 256 KiB of codebooks, which every query re-reads from L2
    const uint32_t qi = blockIdx.x, s = threadIdx.x & 31u, code = blockIdx.y * 8u + (threadIdx.x >> 5);
    const float *q = queries + (size_t)qi * 256 + s * PQ_DSUB;
    const float *cb = codebooks + ((size_t)s * PQ_K + code) * PQ_DSUB;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < PQ_DSUB; ++d) acc += q[d] * cb[d];
    lut[((size_t)qi * PQ_K + code) * PQ_M + s] = acc;
}


// w[pair][k] = scale_l[k] * (Q_l[k] . q) for every (query, probed list) pair; one wave per pair
__global__ void __launch_bounds__(256) lpca_project_kernel(const float *queries, const uint32_t *probe_list, uint64_t n_pairs,
                                                            uint32_t nprobe, const float *basis, const float *lscale, float *w)
{
    const uint64_t pair = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (pair >= n_pairs) return;
    const int lane = threadIdx.x & 63;
    const uint32_t l = probe_list[pair];
    const f32x4 q = reinterpret_cast<const f32x4 *>(queries + (pair / nprobe) * 256)[lane];
    const f32x4 *B = reinterpret_cast<const f32x4 *>(basis + (size_t)l * LP_DIMS * 256);
    float mine = 0.0f;
#pragma unroll 4
    for (int k = 0; k < LP_DIMS; ++k) {
        const f32x4 b = B[k * 64 + lane];
        const float y = wave_sum(q.x * b.x + q.y * b.y + q.z * b.z + q.w * b.w);
        if (lane == k) mine = y * lscale[(size_t)l * LP_DIMS + k];
    }
    if (lane < LP_DIMS) w[pair * LP_DIMS + lane] = mine;
}

// The index ranks by inner products with UNIT rows: the probe score 0.5|c|^2 - q.c and the ADC sums mean "cosine" only for a unit
// query.  Queries of any in-domain length are therefore brought to unit length first (one wave per query; a zero query stays zero);
// the exact select stage at the end re-scores against the query AS GIVEN, like every other path.
__global__ void __launch_bounds__(256) ivf_unit_queries_kernel(const float *queries, uint32_t nq, float *out)
{
    const uint32_t qi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const int lane = threadIdx.x & 63;
    const f32x4 v = reinterpret_cast<const f32x4 *>(queries + (size_t)qi * 256)[lane];
    const float a2 = wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
    const float r = a2 > 0.0f ? __frsqrt_rn(a2) : 0.0f;
    reinterpret_cast<f32x4 *>(out + (size_t)qi * 256)[lane] = f32x4{v.x * r, v.y * r, v.z * r, v.w * r};
}

// ------------------------------------------------------------------ query: ADC scan
struct AdcParams {
    const float *queries;
    const float *lut;          // [nq][256][32]  (kind 0: code-major, see ivf_lut_kernel)
    const float *lw;           // [nq][nprobe][32] per-pair weights of the per-list PCA codes (kind 1), or nullptr
    const uint32_t *probe_list;
    const float *probe_dot;
    uint32_t nprobe;
    uint32_t nq;
    const uint64_t *list_offsets;
    const uint8_t *codes;      // [N][32] in list order
    const uint32_t *ids;       // [N] corpus row of each code
    const float *corpus;       // full-precision rows for the in-kernel re-score
    uint32_t n_seg;            // blocks per (query, probed list): a list is cut into segments of seg_len codes, each
    uint32_t seg_len;          //   with its own shortlist -- the re-scored fraction of a LONG list stays what it is for a short one
    uint32_t shortlist;        // ADC candidates kept per WAVE (<= 64); 4 or 8 waves per (query, list segment)
    uint32_t kp;               // re-scored candidates emitted per (query, list)  (<= 64)
    key_t64 *lists;            // [nq][nprobe][kp]
};

// KIND 0: product quantisation, the query's 256 x 32 LUT staged in LDS (32 KiB: 4 blocks per CU); KIND 1: per-list PCA codes scored
// with 32 block-uniform weights -- no LUT, 2-4 KiB of LDS per block, so the register budget (50 VGPRs) decides the occupancy:
// 8 waves per SIMD instead of 4.
// KIND 0's gathers are CONFLICT-FREE (round 6).  A lane scores its own row: 32 lookups LUT[s][code_s].  With the table laid out
// [s][code] and every lane on the same s at the same time, 32 random codes fell on the 32 banks of ds_read_b32 like balls into bins
// -- 3.5 LDS cycles per lane group instead of 1, 55 % of the kernel's LDS cycles were conflicts and the LDS, not HBM, bounded it
// (0.57 of HBM; profiles/r05_ivf/r05_ivf_pmc_lds.json).  Now the table is [code][s] -- sub-quantiser s lives in bank s -- and lane l
// walks the sub-quantisers in ITS OWN order, s = (t + l) mod 32 at step t: the 32 lanes of a group are on 32 different sub-quantisers,
// hence 32 different banks, at every step, whatever the codes are.  The code bytes stay as the build wrote them (no layout change in
// the index, its files or its append path): each lane rotates its 32-byte record by l mod 32 bytes in registers -- three conditional
// dword stages and one v_alignbyte per dword, 32 VALU instructions per row beside the 96 of the lookups.  (Until round 5 both kinds were one kernel and the unused LUT array halved the resident waves of
// the shipped coding: the re-score stage is random 1 KiB row reads, i.e. latency hidden by waves in flight.)
template <int ADC_THREADS, int KIND>
__global__ void __launch_bounds__(ADC_THREADS) ivf_adc_kernel(AdcParams p)
{
    __shared__ __attribute__((aligned(16))) float s_lut[KIND == 0 ? PQ_M * PQ_K : 4];
    __shared__ key_t64 s_keys[(ADC_THREADS / 64) * 64];
    // XCD-aware block order.  Workgroups go to the 8 XCDs round-robin by their linear id, and each XCD has its own L2: with one grid
    // row per query the P = nprobe x n_seg blocks of a query landed on P different XCDs and every one of them fetched the query's
    // 32 KiB LUT (kind 0) from HBM again -- 262 MB of a 2.4 GB launch (profiles/r05_ivf/, r06_ivf/).  The grid is one line of
    // ceil(nq / 8) x 8 x P blocks: XCD x takes the queries q = 8 j + x, and the P blocks of a query follow each other ON that XCD.
    const uint32_t P = p.nprobe * p.n_seg;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t qi = (slot / P) * 8u + xcd, pblk = slot % P;
    if (qi >= p.nq) return;   // (the last group of eight queries may be short)
    const uint32_t pi = pblk / p.n_seg, seg = pblk % p.n_seg;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kp = (int)p.kp;
    const int ks = (int)p.shortlist;
    key_t64 *out = p.lists + ((size_t)qi * p.nprobe * p.n_seg + pblk) * kp;
    {
        // this block's segment of the probed list; most lists are shorter than n_seg segments: leave an empty list
        const uint32_t l0 = p.probe_list[(size_t)qi * p.nprobe + pi];
        const uint64_t b0 = p.list_offsets[l0], e0 = p.list_offsets[l0 + 1];
        if (b0 + (uint64_t)seg * p.seg_len >= e0) {  // block-uniform
            if ((int)threadIdx.x < kp) out[threadIdx.x] = KEY_PAD;
            return;
        }
    }

    if constexpr (KIND == 0) {
        const f32x4 *lsrc = reinterpret_cast<const f32x4 *>(p.lut + (size_t)qi * PQ_M * PQ_K);
        for (int e = threadIdx.x; e < PQ_M * PQ_K / 4; e += ADC_THREADS) reinterpret_cast<f32x4 *>(s_lut)[e] = lsrc[e];
    }
    // kind 1: the 32 weights of this (query, list) pair, block-uniform (scalar loads)
    float lw[PQ_M];
    if constexpr (KIND == 1) {
        const float *src = p.lw + ((size_t)qi * p.nprobe + pi) * PQ_M;
#pragma unroll
        for (int k = 0; k < PQ_M; ++k) lw[k] = src[k];
    } else {
#pragma unroll
        for (int k = 0; k < PQ_M; ++k) lw[k] = 0.0f;
    }
    float lw_bias = 0.0f;
    if constexpr (KIND == 1) {
#pragma unroll
        for (int k = 0; k < PQ_M; ++k) lw_bias += lw[k];
        lw_bias *= 128.0f;
    }
    const uint32_t rot = (uint32_t)lane & 31u;   // KIND 0: this lane's walk through the sub-quantisers starts at rot
    const uint32_t m16 = (rot & 16u) ? 0xFFFFFFFFu : 0u, m8 = (rot & 8u) ? 0xFFFFFFFFu : 0u, m4 = (rot & 4u) ? 0xFFFFFFFFu : 0u;
    auto bfi = [](uint32_t m, uint32_t a, uint32_t b) -> uint32_t {   // (a & m) | (b & ~m) in ONE instruction, and opaque to the optimiser
        uint32_t d;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(m), "v"(a), "v"(b));
        return d;
    };
    const f32x4 qv = reinterpret_cast<const f32x4 *>(p.queries + (size_t)qi * 256)[lane];
    const float a2 = wave_sum(qv.x * qv.x + qv.y * qv.y + qv.z * qv.z + qv.w * qv.w);
    const bool qz = a2 == 0.0f;
    const float rq = qz ? 0.0f : __frsqrt_rn(a2);
    const uint32_t list = p.probe_list[(size_t)qi * p.nprobe + pi];
    const float base = p.probe_dot[(size_t)qi * p.nprobe + pi];
    const uint64_t begin = p.list_offsets[list] + (uint64_t)seg * p.seg_len;
    const uint64_t list_end = p.list_offsets[list + 1];
    const uint64_t end = seg + 1 == p.n_seg ? list_end : min(list_end, begin + (uint64_t)p.seg_len);  // the last segment takes the rest
    __syncthreads();

    // wave-uniform insert of (cd, cr) into a lane-distributed sorted list of `cap` entries
    auto insert = [&](float cd, uint32_t cr, float &ld, uint32_t &lr, float &thr_d, uint32_t &thr_r, int cap) {
        if (cd < thr_d || (cd == thr_d && cr < thr_r)) {
            const bool less = (ld < cd) || (ld == cd && lr < cr);
            const int pos = __popcll(__ballot(less));
            const float sd = dpp_f<DPP_WAVE_SHR1>(ld);
            const uint32_t sr = dpp_u<DPP_WAVE_SHR1>(lr);
            if (lane > pos) { ld = sd; lr = sr; }
            else if (lane == pos) { ld = cd; lr = cr; }
            thr_d = readlane_f(ld, cap - 1);
            thr_r = (uint32_t)__builtin_amdgcn_readlane((int)lr, cap - 1);
        }
    };

    // ---- stage 1: ADC scan of the list's codes -> the wave's `ks` best approximate candidates (an unordered SET:
    // lane i < n_short ends up holding one of them in (ld, lr)).  A wave takes 64 x ADC_R codes per pass, keeps
    // their ADC distances in registers next to the set carried over from the previous pass, finds the ks-th
    // smallest by bisection on the distance bits (one ballot + scalar popcount per register and step) and
    // compacts the winners through LDS.  (The first version inserted candidates one at a time into a sorted
    // lane-distributed list: ~160 serial inserts per wave at ks = 64 -- that, not the re-score reads, was what
    // bounded this kernel.)
    constexpr int ADC_R = 8;
    float ld = __builtin_inff();       // carried set: lane i < n_carry holds a real entry
    uint32_t lr = 0xFFFFFFFFu;
    key_t64 *s_short = s_keys + wave * 64;  // per-wave compaction scratch (s_keys is reused by the block merge later)
    // (64-code groups are dealt to the waves round-robin, so every wave sees codes from the whole list: lists are in
    // row order and neighbours cluster -- contiguous 512-code chunks per wave cost a point of recall)
    for (uint64_t base_i = begin; base_i < end; base_i += (uint64_t)(ADC_THREADS / 64) * 64 * ADC_R) {
        uint32_t kd[ADC_R + 1], kpos[ADC_R + 1];  // orderable distance bits (0xFFFFFFFF = empty) and list positions
#pragma unroll
        for (int r = 0; r < ADC_R; ++r) {
            const uint64_t i = base_i + ((uint64_t)r * (ADC_THREADS / 64) + wave) * 64 + lane;
            kd[r] = 0xFFFFFFFFu;
            kpos[r] = 0xFFFFFFFFu;
            if (i < end) {
                const uint4 c0 = reinterpret_cast<const uint4 *>(p.codes + i * PQ_M)[0];
                const uint4 c1 = reinterpret_cast<const uint4 *>(p.codes + i * PQ_M)[1];
                const uint32_t w[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                float acc = base;
                if constexpr (KIND == 1) {   // signed bytes times the pair's weights
                    // s = u - 128 with u = s ^ 0x80 as an unsigned byte: one v_cvt_f32_ubyteN per byte instead of a sign-extending
                    // bit-field extract + convert, and 128 x sum(lw) comes off the block-uniform base (lw_bias)
                    acc = base - lw_bias;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const uint32_t x = w[u] ^ 0x80808080u;
                        acc += lw[4 * u + 0] * (float)(x & 0xFF);
                        acc += lw[4 * u + 1] * (float)((x >> 8) & 0xFF);
                        acc += lw[4 * u + 2] * (float)((x >> 16) & 0xFF);
                        acc += lw[4 * u + 3] * (float)(x >> 24);
                    }
                } else {
                    // the record rotated left by rot = lane % 32 bytes: byte t of x[] is the code of sub-quantiser (t + rot) % 32
                    // (bit selects through lane masks, one v_bfi_b32 each, as inline asm: written as `rot & 16 ? a : b` clang folds the three stages into ONE
                    // dynamically indexed pick per dword -- seven compare + select pairs each, 1400 of them per pass, 0.36 of HBM)
                    uint32_t x[8], y[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) y[u] = bfi(m16, w[(u + 4) & 7], w[u]);
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = bfi(m8, y[(u + 2) & 7], y[u]);
#pragma unroll
                    for (int u = 0; u < 8; ++u) y[u] = bfi(m4, x[(u + 1) & 7], x[u]);
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = __builtin_amdgcn_alignbyte(y[(u + 1) & 7], y[u], rot & 3);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        acc += s_lut[((x[u] & 0xFF) << 5) | ((4 * u + 0 + rot) & 31)];
                        acc += s_lut[(((x[u] >> 8) & 0xFF) << 5) | ((4 * u + 1 + rot) & 31)];
                        acc += s_lut[(((x[u] >> 16) & 0xFF) << 5) | ((4 * u + 2 + rot) & 31)];
                        acc += s_lut[((x[u] >> 24) << 5) | ((4 * u + 3 + rot) & 31)];
                    }
                }
                const float d = fmaxf(1.0f - acc * rq, 0.0f);  // rows are unit-norm (model2vec output), zero rows score ~0
                if (acc == acc) {                                // a NaN score never becomes a candidate
                    kd[r] = min(__float_as_uint(d), 0xFFFFFFFEu);  // d >= 0: the bit pattern orders like the value
                    kpos[r] = (uint32_t)i;                         // position in list order (codes / ids / int8 rows share it)
                }
            }
        }
        kd[ADC_R] = lr != 0xFFFFFFFFu ? __float_as_uint(ld) : 0xFFFFFFFFu;  // (ld keeps only the top 16 bits: enough here)
        kpos[ADC_R] = lr;
        // (wave-wide counts through ballots: the compare writes a lane mask to SGPRs and s_bcnt1 counts it on the scalar unit --
        // 9 VALU instructions per bisection step instead of 9 + 9 + an 11-instruction lane reduction)
        uint32_t total = 0;
#pragma unroll
        for (int r = 0; r <= ADC_R; ++r) total += (uint32_t)__popcll(__ballot(kd[r] != 0xFFFFFFFFu));
        // The ADC distance is itself an approximation (error ~1e-2): its top 16 bits (relative step 2^-8 of the value)
        // are all the selection needs, which halves the bisection; ties in that bucket go by scan order.
#pragma unroll
        for (int r = 0; r <= ADC_R; ++r) kd[r] = kd[r] == 0xFFFFFFFFu ? 0xFFFFFFFFu : (kd[r] >> 16);
        uint32_t T = 0xFFFFFFFEu, need_eq = 0xFFFFFFFFu;  // winners: kd < T, plus the first need_eq entries with kd == T
        if (total > (uint32_t)ks) {
            uint32_t lo = 0u, hi = 0xFFFFu;               // smallest T with #(kd <= T) >= ks
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                uint32_t cnt = 0;
#pragma unroll
                for (int r = 0; r <= ADC_R; ++r) cnt += (uint32_t)__popcll(__ballot(kd[r] <= mid));
                if (cnt >= (uint32_t)ks) hi = mid; else lo = mid + 1u;
            }
            T = lo;
            uint32_t n_lt = 0;
#pragma unroll
            for (int r = 0; r <= ADC_R; ++r) n_lt += (uint32_t)__popcll(__ballot(kd[r] < T));
            need_eq = (uint32_t)ks - n_lt;
        }
        // compaction: winners take consecutive LDS slots, then lane i reads slot i
        uint32_t n_out = 0, n_eq_seen = 0;
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int r = 0; r <= ADC_R; ++r) {
            const bool valid = kd[r] != 0xFFFFFFFFu;
            const bool lt = valid && kd[r] < T;
            const bool eq = valid && kd[r] == T;
            const unsigned long long m_eq = __ballot(eq);
            const bool eq_win = eq && (n_eq_seen + (uint32_t)__popcll(m_eq & below)) < need_eq;
            const unsigned long long m_win = __ballot(lt || eq_win);
            if (lt || eq_win) s_short[n_out + (uint32_t)__popcll(m_win & below)] = ((key_t64)kd[r] << 32) | kpos[r];
            n_out += (uint32_t)__popcll(m_win);
            n_eq_seen += (uint32_t)__popcll(m_eq);
        }
        __builtin_amdgcn_wave_barrier();
        const key_t64 mine = (uint32_t)lane < n_out ? reinterpret_cast<volatile key_t64 *>(s_short)[lane] : KEY_PAD;
        __builtin_amdgcn_wave_barrier();
        ld = mine != KEY_PAD ? __uint_as_float((uint32_t)(mine >> 32) << 16) : __builtin_inff();
        lr = mine != KEY_PAD ? (uint32_t)(mine & 0xFFFFFFFFull) : 0xFFFFFFFFu;
    }

    // (An int8 refinement stage between the two -- a 260 B/row copy of the rows ranking the shortlist so that only a few
    // candidates need their 1 KiB row -- was built in round 1, measured at +6 % queries/s for a 7x larger index, kept opt-in
    // for two rounds and removed in round 3.)
    const int n_short = __popcll(__ballot(lr != 0xFFFFFFFFu));  // the set sits in lanes 0..n_short-1
    unsigned long long go = n_short >= 64 ? ~0ull : ((1ull << n_short) - 1ull);  // lanes whose candidate is re-scored

    // ---- stage 2: re-score the survivors with the full-precision rows (coalesced 1 KiB loads, f32),
    //      keep the kp best; the select stage then recomputes those exactly in f64
    const uint32_t my_row = (lane < n_short && ((go >> lane) & 1ull)) ? p.ids[lr] : 0xFFFFFFFFu;  // one gather, before the loop
    float ld2 = __builtin_inff();
    uint32_t lr2 = 0xFFFFFFFFu;
    float thr2_d = __builtin_inff();
    uint32_t thr2_r = 0xFFFFFFFFu;
    // Rows in flight per wave: the reads are random 1 KiB rows, i.e. latency, and what hides it is rows in flight per CU.  The PQ
    // kind's 32 KiB LUT keeps it at 4 waves per SIMD where the per-list PCA kind runs 7, so its waves keep EIGHT rows in flight
    // instead of four (round 6: the re-scored rows are 60 % of this kernel's bytes -- 128 KiB per block against 88 KiB of codes --
    // and their latency, not the LUT gathers, was what held the PQ kind at 0.57 of HBM).
    constexpr int RS = KIND == 0 ? 8 : 4;
    while (go) {
        f32x4 c[RS];
        uint32_t rr[RS];
        bool ok[RS];
#pragma unroll
        for (int u = 0; u < RS; ++u) {
            ok[u] = go != 0ull;
            const int src = ok[u] ? __ffsll((long long)go) - 1 : 0;
            if (ok[u]) go &= go - 1;
            rr[u] = (uint32_t)__builtin_amdgcn_readlane((int)my_row, src);
            c[u] = reinterpret_cast<const f32x4 *>(p.corpus + (uint64_t)(ok[u] ? rr[u] : 0u) * 256)[lane];
        }
        // four rows' norms and dot products reduced together (device_utils.h wave_sum4: lane l ends with the sum of row l % 4)
#pragma unroll
        for (int g4 = 0; g4 < RS; g4 += 4) {
            float pb[4], pa[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 cu = c[g4 + u];
                pb[u] = cu.x * cu.x + cu.y * cu.y + cu.z * cu.z + cu.w * cu.w;
                pa[u] = cu.x * qv.x + cu.y * qv.y + cu.z * qv.z + cu.w * qv.w;
            }
            const float b2s = wave_sum4(pb[0], pb[1], pb[2], pb[3], lane);
            const float abs4 = wave_sum4(pa[0], pa[1], pa[2], pa[3], lane);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float b2 = readlane_f(b2s, u), ab = readlane_f(abs4, u);
                if (ok[g4 + u]) insert(dist_f32(ab, b2, rq, qz), rr[g4 + u], ld2, lr2, thr2_d, thr2_r, kp);
            }
        }
    }

    // block merge of the wave lists (rank by counting), as in K2
    s_keys[wave * 64 + lane] = (lane < kp && lr2 != 0xFFFFFFFFu) ? make_key(ld2, lr2) : KEY_PAD;
    if ((int)threadIdx.x < kp) out[threadIdx.x] = KEY_PAD;
    __syncthreads();
    const key_t64 mine = s_keys[wave * 64 + lane];
    if (mine != KEY_PAD) {
        int rank = 0;
        for (int w = 0; w < ADC_THREADS / 64; ++w)
            for (int i = 0; i < kp; ++i) rank += (s_keys[w * 64 + i] < mine) ? 1 : 0;
        if (rank < kp) out[rank] = mine;
    }
}

}  // namespace smt

using namespace smt;


extern "C" {

static int ivfpq_search_core(smt_ivfpq *ix, const float *queries, bool queries_on_device, uint32_t nq, uint32_t top_k, uint32_t nprobe,
                             uint32_t rerank, uint64_t row_base, uint64_t *d_or_user, double *d_od_user, uint64_t *d_oc_user,
                             uint64_t **d_or_out, size_t *out_bytes_contig, uint64_t out_stride = 0, smt::Delivery *deliver = nullptr)
{
    smt_ctx *ctx = ix->corpus->ctx;
    SMT_REQUIRE(ix->corpus->rows >= ix->n_rows, "the corpus shrank after the index was built: rebuild");
    SMT_REQUIRE(nprobe >= 1 && nprobe <= ix->nlist && nprobe <= 512, "nprobe must be in [1, min(nlist, 512)]");
    SMT_REQUIRE(top_k <= 56, "top_k must be <= 56 for the IVF-PQ path");
    if (rerank == 0) rerank = 512;
    SMT_REQUIRE(rerank >= 4 && rerank <= 512, "rerank (full-precision re-scored ADC candidates per probed list) must be in [4, 512]");
    // A list longer than ADC_SEGMENT codes is scanned by several blocks, each with its own shortlist of `rerank`
    // candidates (config 5's 100 M rows over 4096 lists: 24 k codes per list -- one shortlist of 512 would re-score
    // 2 % of them and recall@10 drops to 0.75); the select stage takes at most 512 lists per query.
    constexpr uint64_t ADC_SEGMENT = 8192;
    // sized by the TYPICAL list (1.5 x the mean), not the longest: a block that finds its segment empty still costs a
    // launch slot (+0.3 ms per 1000 queries when every list got a second, almost always empty, segment); the last
    // segment of an unusually long list simply takes the rest
    const uint64_t typical = (ix->n_rows / std::max<uint32_t>(ix->nlist, 1u)) * 3 / 2;
    uint32_t n_seg = (uint32_t)std::max<uint64_t>(1, (typical + ADC_SEGMENT - 1) / ADC_SEGMENT);
    // ... and when that limit takes segments away (nprobe 128 over 100 M rows: 4 instead of 5), the typical list is cut into EQUAL
    // longer segments -- with ADC_SEGMENT kept, the last one took a double share behind one shortlist and recall@10 FELL with
    // nprobe (0.9795 at 32, 0.9708 at 128)
    const uint32_t want_seg = n_seg;
    n_seg = std::min<uint32_t>(n_seg, std::max<uint32_t>(1u, 512u / nprobe));
    const uint32_t seg_len = n_seg < want_seg ? (uint32_t)(((typical + n_seg - 1) / n_seg + 255) & ~(uint64_t)255) : (uint32_t)ADC_SEGMENT;
    // (... each with a shortlist longer by the same factor: the re-scored fraction of a list does not depend on nprobe)
    if (n_seg < want_seg) rerank = std::min<uint32_t>(512u, (rerank * want_seg + n_seg - 1) / n_seg);

    // waves per (query, list segment) block.  (PQ, kind 0, with 8-wave blocks -- two blocks' worth of waves sharing one 32 KiB LUT --
    // measured +4 % queries/s for -0.5 point of recall@10 at rerank 128: sixteen candidates per wave are too few.  Not taken.)
    const int adc_waves = rerank > 256 ? 8 : 4;
    const uint32_t shortlist = (rerank + adc_waves - 1) / adc_waves;  // per wave
    const uint32_t kp = top_k + 8;                // re-scored candidates handed to the exact select stage
    // every temporary lives in the context's scratch (no hipMalloc/hipFree per call), results come back through
    // the pinned staging buffer
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_q = 0, b_q = 2 * al((size_t)nq * 256 * 4);   // [the queries as given (host form) | their unit-length copies]
    const size_t o_pl = o_q + b_q, b_pl = al((size_t)nq * nprobe * 4);
    const size_t o_pd = o_pl + b_pl, b_pd = b_pl;
    const size_t o_lut = o_pd + b_pd, b_lut = al((size_t)nq * PQ_M * PQ_K * 4);
    const size_t o_lists = o_lut + b_lut, b_lists = al((size_t)nq * nprobe * n_seg * kp * 8);
    const size_t o_or = o_lists + b_lists, b_or = (size_t)nq * top_k * 8;   // rows | dist | counts contiguous: one D2H
    const size_t o_od = o_or + b_or, b_od = b_or;
    const size_t o_oc = o_od + b_od, b_oc = al((size_t)nq * 8);
    const size_t o_sc = o_oc + b_oc, b_sc = al((size_t)nq * ix->nlist * 4);
    const size_t o_lw = o_sc + b_sc, b_lw = ix->kind == 1 ? al((size_t)nq * nprobe * PQ_M * 4) : 0;
    int rc = smt::ensure_scratch(ctx, o_lw + b_lw);
    if (rc) return rc;
    char *base = reinterpret_cast<char *>(ctx->d_scratch);
    const float *d_q = queries;
    if (!queries_on_device) {
        IVF_HIP(hipMemcpyAsync(base + o_q, queries, (size_t)nq * 256 * 4, hipMemcpyHostToDevice, ctx->stream));
        d_q = reinterpret_cast<const float *>(base + o_q);
    }
    const float *d_q_given = d_q;   // (the exact select re-scores against these)
    {
        float *d_unit = reinterpret_cast<float *>(base + o_q + b_q / 2);
        hipLaunchKernelGGL(ivf_unit_queries_kernel, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, d_q, nq, d_unit);
        d_q = d_unit;
    }

    if (!(ctx->attr_done & ATTR_IVF_SCORE)) {
        IVF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ivf_score_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->attr_done |= ATTR_IVF_SCORE;
    }
    ProbeParams pp;
    pp.queries = d_q;
    pp.centroids = ix->d_centroids;
    pp.cnorm_half = ix->d_cnorm_half;
    pp.nlist = ix->nlist;
    pp.nprobe = nprobe;
    pp.probe_list = reinterpret_cast<uint32_t *>(base + o_pl);
    pp.probe_dot = reinterpret_cast<float *>(base + o_pd);
    float *d_scores = reinterpret_cast<float *>(base + o_sc);
    prof_begin(ctx, "ivf_probe");
    hipLaunchKernelGGL(ivf_score_kernel, dim3(ix->nlist / QT_ROWS), dim3(GEMM_THREADS), (size_t)QT_F4 * 16 + 64, ctx->stream, d_q, nq,
                       ix->d_centroids, ix->d_cnorm_half, ix->nlist, d_scores);
    hipLaunchKernelGGL(ivf_probe_select_kernel, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, pp, d_scores, nq);
    if (ix->kind == 1) {
        const uint64_t n_pairs = (uint64_t)nq * nprobe;
        hipLaunchKernelGGL(lpca_project_kernel, dim3((unsigned)((n_pairs * 64 + 255) / 256)), dim3(256), 0, ctx->stream, d_q, pp.probe_list,
                           n_pairs, nprobe, ix->d_basis, ix->d_lscale, reinterpret_cast<float *>(base + o_lw));
    } else {
        hipLaunchKernelGGL(ivf_lut_kernel, dim3(nq, PQ_K / 8), dim3(256), 0, ctx->stream, d_q, ix->d_codebooks, reinterpret_cast<float *>(base + o_lut));
    }
    prof_end(ctx, "ivf_probe");
    AdcParams ap;
    ap.queries = d_q;
    ap.lut = reinterpret_cast<float *>(base + o_lut);
    ap.lw = ix->kind == 1 ? reinterpret_cast<const float *>(base + o_lw) : nullptr;
    ap.probe_list = pp.probe_list;
    ap.probe_dot = pp.probe_dot;
    ap.nprobe = nprobe;
    ap.nq = nq;
    ap.list_offsets = ix->d_offsets;
    ap.codes = ix->d_codes;
    ap.ids = ix->d_ids;
    ap.corpus = ix->corpus->d_rows;
    ap.n_seg = n_seg;
    ap.seg_len = seg_len ? seg_len : 512;
    ap.shortlist = shortlist;
    ap.kp = kp;
    ap.lists = reinterpret_cast<key_t64 *>(base + o_lists);
    prof_begin(ctx, "ivf_adc");
    const dim3 adc_grid(((nq + 7) / 8) * 8 * nprobe * n_seg);   // (one line: see the XCD-aware order in the kernel)
    if (ix->kind == 1) {
        if (adc_waves == 8) hipLaunchKernelGGL((ivf_adc_kernel<512, 1>), adc_grid, dim3(512), 0, ctx->stream, ap);
        else hipLaunchKernelGGL((ivf_adc_kernel<256, 1>), adc_grid, dim3(256), 0, ctx->stream, ap);
    } else {
        if (adc_waves == 8) hipLaunchKernelGGL((ivf_adc_kernel<512, 0>), adc_grid, dim3(512), 0, ctx->stream, ap);
        else hipLaunchKernelGGL((ivf_adc_kernel<256, 0>), adc_grid, dim3(256), 0, ctx->stream, ap);
    }
    prof_end(ctx, "ivf_adc");
    IVF_HIP(hipGetLastError());
    uint64_t *d_or = d_or_user ? d_or_user : reinterpret_cast<uint64_t *>(base + o_or);
    double *d_od = d_od_user ? d_od_user : reinterpret_cast<double *>(base + o_od);
    uint64_t *d_oc = d_or_user ? d_oc_user : reinterpret_cast<uint64_t *>(base + o_oc);
    SelectArgs sel;  // no exactness certificate: the index is approximate by contract (f32_err = 0)
    sel.corpus = ix->corpus->d_rows;
    sel.queries = d_q_given;
    sel.nq = nq;
    sel.lists = ap.lists;
    sel.n_lists = nprobe * n_seg;
    sel.kp = kp;
    sel.list_stride = (uint64_t)nprobe * n_seg * kp;
    sel.k_out = top_k;
    sel.row_base = row_base;
    sel.out_rows = d_or;
    sel.out_dist = d_od;
    sel.out_counts = d_oc;
    sel.out_stride = out_stride;
    if (deliver) {   // (host form, small answer: the select kernel carries [rows | distances | counts] home -- common.h Delivery)
        deliver->dev_out = reinterpret_cast<const unsigned long long *>(d_or);
        deliver->n_words = (uint32_t)((b_or + b_od + (size_t)nq * 8) / 8);
        sel.deliver = deliver;
    }
    rc = launch_select(ctx, sel);
    if (rc) return rc;
    if (d_or_out) *d_or_out = d_or;
    if (out_bytes_contig) *out_bytes_contig = b_or + b_od + (size_t)nq * 8;
    return SMT_OK;
}

int smt_ivfpq_search(smt_ivfpq *ix, const float *queries, uint32_t nq, uint32_t top_k, uint32_t nprobe, uint32_t rerank,
                     uint64_t row_base, uint64_t *out_rows, double *out_dist, uint64_t *out_counts, uint64_t out_cap)
try {
    SMT_REQUIRE(ix != nullptr, "index");
    SMT_REQUIRE(nq == 0 || (queries && out_rows && out_dist && out_counts), "null argument");
    smt_ctx *ctx = ix->corpus->ctx;
    IVF_HIP(hipSetDevice(ctx->device));
    { int rc_drain = smt::drain_async(ctx); if (rc_drain) return rc_drain; }
    if (nq == 0) return SMT_OK;
    for (uint32_t q = 0; q < nq; ++q) out_counts[q] = 0;
    if (int rcq = smt::require_queries_domain_host(queries, nq, "smt_ivfpq_search")) return rcq;   // (domain.hip)
    if (top_k == 0) return SMT_OK;
    uint64_t *d_or = nullptr;
    size_t out_bytes = 0;
    const size_t b_or = (size_t)nq * top_k * 8, b_od = b_or;
    // a small answer is delivered by the select kernel (as in smt_search: search.cpp), a large one copied and waited for
    const bool direct = ctx->tune.direct_delivery != 0 && nq <= 32 && b_or + b_od + (size_t)nq * 8 <= 8192;
    int rc = smt::ensure_pinned(ctx, 64 + b_or + b_od + (size_t)nq * 8);
    if (rc) return rc;
    char *h_ans = reinterpret_cast<char *>(ctx->h_pinned) + 64;
    volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(ctx->h_pinned);
    smt::Delivery dl;
    if (direct) {
        dl.host_out = reinterpret_cast<unsigned long long *>(h_ans);
        dl.host_flag = const_cast<unsigned long long *>(flag);
        dl.seq = ++ctx->deliver_seq;
        dl.done = ctx->d_status + 4;
        *flag = 0;
    }
    rc = ivfpq_search_core(ix, queries, false, nq, top_k, nprobe, rerank, row_base, nullptr, nullptr, nullptr, &d_or, &out_bytes, 0,
                           direct ? &dl : nullptr);
    if (rc) return rc;
    if (direct) {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned long long got = 0;
        for (unsigned spins = 1; (got = *flag) == 0; ++spins) {
            if ((spins & 63) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
                IVF_HIP(hipStreamSynchronize(ctx->stream));
                got = *flag;
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (got != dl.seq) {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipMemsetAsync(ctx->d_status + 4, 0, sizeof(unsigned long long), ctx->stream);
            smt::set_error("the select kernel did not deliver its answer (completion word %llu, expected %llu)", got, dl.seq);
            return SMT_E_HIP;
        }
        ++ctx->deliveries;
    } else {
        IVF_HIP(hipMemcpyAsync(h_ans, d_or, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        IVF_HIP(hipStreamSynchronize(ctx->stream));
    }
    const uint64_t *h_rows = reinterpret_cast<const uint64_t *>(h_ans);
    const double *h_dist = reinterpret_cast<const double *>(h_ans + b_or);
    const uint64_t *h_cnt = reinterpret_cast<const uint64_t *>(h_ans + b_or + b_od);
    bool truncated = false;
    for (uint32_t q = 0; q < nq; ++q) {
        out_counts[q] = h_cnt[q];
        const uint64_t w = std::min<uint64_t>(h_cnt[q], out_cap);
        if (h_cnt[q] > out_cap) truncated = true;
        for (uint64_t i = 0; i < w; ++i) {
            out_rows[(size_t)q * out_cap + i] = h_rows[(size_t)q * top_k + i];
            out_dist[(size_t)q * out_cap + i] = h_dist[(size_t)q * top_k + i];
        }
    }
    if (truncated) { smt::set_error("out_cap smaller than the number of hits"); return SMT_E_TRUNCATED; }
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_ivfpq_search_device(smt_ivfpq *ix, const float *queries_dev, uint32_t nq, uint32_t top_k, uint32_t nprobe, uint32_t rerank,
                            uint64_t row_base, uint64_t *out_rows_dev, double *out_dist_dev)
try {
    SMT_REQUIRE(ix != nullptr, "index");
    SMT_REQUIRE(nq == 0 || (queries_dev && out_rows_dev && out_dist_dev), "null argument");
    SMT_REQUIRE(top_k >= 1, "top_k");
    smt_ctx *ctx = ix->corpus->ctx;
    IVF_HIP(hipSetDevice(ctx->device));
    { int rc_drain = smt::drain_async(ctx); if (rc_drain) return rc_drain; }
    if (nq == 0) return SMT_OK;
    return ivfpq_search_core(ix, queries_dev, true, nq, top_k, nprobe, rerank, row_base, out_rows_dev, out_dist_dev, nullptr, nullptr, nullptr);
} catch (...) { return smt::api_catch(); }

}  // extern "C"

// one shard's answer in the packed exchange layout of group.cpp: [nq][2][top_k] words (global rows | f64 bits)
int smt::ivfpq_search_packed(smt_ivfpq *ix, const float *queries_dev, uint32_t nq, uint32_t top_k, uint32_t nprobe, uint32_t rerank,
                             uint64_t row_base, uint64_t *packed_dev)
{
    SMT_REQUIRE(ix && queries_dev && packed_dev, "null argument");
    smt_ctx *ctx = ix->corpus->ctx;
    IVF_HIP(hipSetDevice(ctx->device));
    { int rc_drain = smt::drain_async(ctx); if (rc_drain) return rc_drain; }
    return ivfpq_search_core(ix, queries_dev, true, nq, top_k, nprobe, rerank, row_base, packed_dev,
                             reinterpret_cast<double *>(packed_dev + top_k), nullptr, nullptr, nullptr, (uint64_t)2 * top_k);
}