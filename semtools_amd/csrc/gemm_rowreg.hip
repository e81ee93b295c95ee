// gemm_rowreg.hip -- gemm_rowreg_kernel<MODE, IMG>: the default K3 kernel of every unfiltered batch, the nomination
// buffer in LDS, and pack_image_kernel, which writes the corpus' fp16 operand image the IMG instantiations read.
#include "gemm.h"

namespace smt {

// ---- bf16 x 3, every unfiltered batch size: row tiles arrive in REGISTERS with coalesced loads and are transposed
// into the MFMA operand layout through a small wave-private LDS buffer.
//
// Why: the MFMA A layout wants lane j <-> row j, and loading a row tile directly in that layout means 32 instructions
// that each touch 16- or 32-byte pieces of 32 different rows.  Measured on MI355X (tools/micro/row_load_patterns.hip,
// 8 waves per CU, 10 M rows): such loads deliver 3.2-3.6 TB/s whatever the sweep behind them; instructions that read
// 128-byte runs (8 lanes per row, 8 rows per instruction) deliver 7.0-7.2 TB/s, also with 192 MFMAs per tile behind
// them.  So: 32 coalesced loads per tile -> 1/|row| scaling and the bf16 hi/lo split ONCE per tile in that layout ->
// per 32-dim slice the packed words go through a 2.5 KiB LDS buffer (80-byte row stride: conflict-free b128 reads)
// and come back as the operand quads Ah/Al[16], which then serve the whole sweep over the query tiles.  The query
// tiles are the split image (split_queries_kernel) in LDS: resident up to 4 tiles, else streamed in pairs like
// gemm_level_kernel.  Thresholds and 1/|q| are read from global memory (kept per level by level_select_kernel), so
// the LDS footprint does not depend on the batch size.
// MODE 0: bf16 x 3.  MODE 1: f16 x 2 nomination (mfma_tile.h) -- the rows carry ONE fp16 operand (64 VGPRs), the transpose
// moves half the words, a K-step is two MFMAs (row x query-hi, row x query-lo).  MODE 2: f16 x 1 -- the query's lo part is
// dropped too: ONE MFMA and ONE B quad per K-step, half the MFMA and half the LDS operand traffic of f16 x 2 for a
// certificate band of 2^-10 instead of 2^-11 (common.h F32_ERR_F16X1): the large-batch mode, where the MFMA pipe -- at the
// clock the part sustains under this load -- is the bound and the only lever left is fewer MFMAs per useful flop.
#if defined(SMT_RR_EXP) && (SMT_RR_EXP & 256)
#define CB_STAMP(id) do { if (dbg) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) dbg[dbg_n] = (t_ << 8) | (unsigned)(id); ++dbg_n; } } while (0)
#define CB_DBG_PARAMS , unsigned long long *dbg, int &dbg_n
#else
#define CB_STAMP(id) do { } while (0)
#define CB_DBG_PARAMS
#endif
// the wave's nomination buffer (its transpose buffer, idle during a sweep): keys at cb, queries behind them; n_buf is wave-uniform
__device__ __forceinline__ void append_candidates_lds(const f32x16 &acc, unsigned zero16, unsigned valid16, uint32_t q, float thr, float rq,
                                                      uint64_t row0, int h, int lane, unsigned char *cb, uint32_t &n_buf,
                                                      key_t64 *cand, unsigned int *counts CB_DBG_PARAMS)
{
    // Every VALU instruction of an epilogue competes with the MFMA stream of the SIMD's other wave (~12 cycles apiece there), and
    // while it runs the block's other waves may be standing at the ring barrier.  So, in order of frequency:
    //  (1) nothing to nominate (~90 % of the products at 1000 x 10 M): eight v_max3, one compare, one branch;
    //  (2) nominations of nonzero queries in a full tile: one v_cmp per accumulator register, its lane mask in SGPRs -- scalar
    //      tests skip the registers without a nomination; a register with some: slot = count + mbcnt, two LDS writes;
    //  (3) zero queries with a reachable threshold, tiles that hang over the end of the corpus: the general mask, bit by bit.
    const bool zq = rq == 0.0f;    // a zero query's slot carries tau itself (score_threshold): padding has tau < 0
    {
        float mx = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
        for (int r = 3; r + 1 < 16; r += 2) mx = fmaxf(fmaxf(mx, acc[r]), acc[r + 1]);
        mx = fmaxf(mx, acc[15]);
        if (!__builtin_amdgcn_ballot_w64(zq ? thr >= 0.0f : mx >= thr)) return;
    }
    CB_STAMP(11);
    key_t64 *keys = reinterpret_cast<key_t64 *>(cb);
    uint32_t *qs = reinterpret_cast<uint32_t *>(cb + RR_CB_CAP * 8);
    if (__builtin_amdgcn_ballot_w64((zq && thr >= 0.0f) || valid16 != 0xffffu)) {
        unsigned pass = 0;
        if (!zq) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (acc[r] >= thr) pass |= 1u << r;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (nomination_dist(acc, r, zero16, rq) <= thr) pass |= 1u << r;
        }
        pass &= valid16;
        const unsigned long long lanes = __builtin_amdgcn_ballot_w64(pass != 0);
        if (!lanes) return;
        const uint32_t mine = (uint32_t)__popc(pass);
        uint32_t slot = 0, run = n_buf;
        for (unsigned long long m = lanes; m; m &= m - 1) {
            const int l = __builtin_ctzll(m);
            if (lane == l) slot = run;
            run += (uint32_t)__builtin_amdgcn_readlane((int)mine, l);
        }
        if (run > (uint32_t)RR_CB_CAP) {   // wave-uniform: no room: this tile goes straight to the lists
            append_direct(acc, pass, zero16, q, rq, row0, h, cand, counts);
            return;
        }
        n_buf = run;
        unsigned todo = pass;
        while (__builtin_amdgcn_ballot_w64(todo != 0)) {
            const int r = todo ? __builtin_ctz(todo) : 0;
            const float a = acc_select(acc, r);
            if (todo) {
                const float d = zq ? ((zero16 >> r) & 1u ? 0.0f : 1.0f) : fmaxf(1.0f - a * rq, 0.0f);
                keys[slot] = make_key(d, (uint32_t)(row0 + acc_row(r, h)));
                qs[slot] = q;
                ++slot;
                todo &= todo - 1;
            }
        }
        return;
    }
    const unsigned long long nonzero_q = __builtin_amdgcn_ballot_w64(!zq);
    const uint32_t lo = (uint32_t)lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned long long mr = __builtin_amdgcn_ballot_w64(acc[r] >= thr) & nonzero_q;
        if (__builtin_expect(mr != 0, 0)) {
            const uint32_t n = (uint32_t)__builtin_popcountll(mr);
            const bool mine = (mr >> lo) & 1ull;
            const key_t64 key = make_key(fmaxf(1.0f - acc[r] * rq, 0.0f), (uint32_t)(row0 + acc_row(r, h)));
            if (n_buf + n <= (uint32_t)RR_CB_CAP) {
                if (mine) {
                    const uint32_t slot = n_buf + __builtin_amdgcn_mbcnt_hi((uint32_t)(mr >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mr, 0u));
                    keys[slot] = key;
                    qs[slot] = q;
                }
                n_buf += n;
            } else if (mine) {   // no room (a loose threshold): straight to the list
                const unsigned slot = atomicAdd(&counts[q], 1u);
                if (slot < CAND_CAP) cand[(size_t)q * CAND_CAP + slot] = key;
            }
        }
    }
    CB_STAMP(13);
}
__device__ __forceinline__ void flush_candidates_lds(int lane, const unsigned char *cb, uint32_t &n_buf, key_t64 *cand, unsigned int *counts)
{
    const key_t64 *keys = reinterpret_cast<const key_t64 *>(cb);
    const uint32_t *qs = reinterpret_cast<const uint32_t *>(cb + RR_CB_CAP * 8);
    for (uint32_t i = (uint32_t)lane; i < n_buf; i += 64) {
        const uint32_t q = qs[i];
        const unsigned slot = atomicAdd(&counts[q], 1u);
        if (slot < CAND_CAP) cand[(size_t)q * CAND_CAP + slot] = keys[i];
    }
    n_buf = 0;
}

// THE CORPUS' fp16 OPERAND IMAGE (smt_corpus::image, api.cpp): per 32-row tile the sixteen operand quads gemm_rowreg_kernel's fp16
// modes build in their row phase -- unit row x 2^10, fp16, quad (K-step m, lane l = 32 h + j) at 16 (64 m + l) bytes of the
// tile's 16 KiB -- written ONCE per row by this kernel with the same arithmetic in the same order (the tests compare the
// nominations of both forms bit for bit), plus the tile's zero-row mask.  One wave per tile.
__global__ void __launch_bounds__(256) pack_image_kernel(const float *corpus, uint64_t n_rows, uint64_t first_tile, uint64_t n_tiles,
                                                         uint32_t *image, uint32_t *image_zero)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_t[4 * RR_TBUF];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const uint64_t t = (uint64_t)blockIdx.x * 4 + wave;
    if (t >= n_tiles) return;
    const uint64_t tile = first_tile + t, row0 = tile * 32;
    unsigned char *tbuf = s_t + wave * RR_TBUF;
    const uint32_t t_wr = (uint32_t)((lane >> 3) * RR_TROW + (lane & 7) * 8);
    const uint32_t t_rd = (uint32_t)(j * RR_TROW + h * 16);
    f32x4 R[32];
    {
        const f32x4 *base = reinterpret_cast<const f32x4 *>(corpus) + (lane & 7);
        uint64_t rowv[4];
        bool inside[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t r = row0 + 8 * u + (lane >> 3);
            inside[u] = r < n_rows;
            rowv[u] = inside[u] ? r : 0;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            R[i] = __builtin_nontemporal_load(base + rowv[i & 3] * 64 + 8 * (i >> 2));
            if (!inside[i & 3]) R[i] = (f32x4){0.f, 0.f, 0.f, 0.f};   // rows past the end: zero rows (masked by valid16 in the sweep)
        }
    }
    float rb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float part = 0.0f;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            const f32x4 v = R[4 * sl + u];
            part += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 4);
        rb[u] = part == 0.0f ? 0.0f : __frsqrt_rn(part);
    }
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x4 *out = reinterpret_cast<u32x4 *>(image) + tile * 1024 + lane;
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 v = R[4 * sl + u] * (rb[u] * F16X2_ROW_SCALE);
            *reinterpret_cast<u32x2 *>(tbuf + t_wr + u * 8 * RR_TROW) = (u32x2){f16_pack2(v.x, v.y), f16_pack2(v.z, v.w)};
        }
        out[(2 * sl) * 64] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd);
        out[(2 * sl + 1) * 64] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd + 32);
    }
    uint32_t zm = 0;   // bit 8u + i: tile row 8u + i (its 1/|row| sits in lanes 8i .. 8i + 7 of rb[u])
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned long long z = __ballot(rb[u] == 0.0f);
#pragma unroll
        for (int i = 0; i < 8; ++i) zm |= (uint32_t)((z >> (8 * i)) & 1ull) << (8 * u + i);
    }
    if (lane == 0) image_zero[tile] = zm;
}

#ifndef SMT_RR_EXP
#define SMT_RR_EXP 0
#endif
#if (SMT_RR_EXP & 256)
// wave timeline of block 40, second step, main level (level_tiles large): (s_memtime << 8) | id, 1024 stamps per wave
#define RR_STAMP(id) do { if (tracing && n_stamp < 1024) { const unsigned long long t_ = __builtin_readcyclecounter(); \
        if (lane == 0) p.stamps[wave * 1024 + n_stamp] = (t_ << 8) | (unsigned)(id); ++n_stamp; } } while (0)
#else
#define RR_STAMP(id) do { } while (0)
#endif
template <int MODE, bool IMG = false>
__global__ void __launch_bounds__(RR_THREADS, 2) gemm_rowreg_kernel(GemmParams p)
{
    static_assert(!IMG || MODE >= 1, "the operand image holds fp16 rows");
    constexpr bool F16X2 = MODE >= 1;     // fp16 row operand (MODE 1 and 2)
    constexpr bool F16X1 = MODE == 2;     // ... and a single fp16 query operand
    // (plain constants, not RrGeom<MODE>::X inside the lambdas below: hipcc 7.2 silently drops the HOST-side instantiation of a
    // kernel template whose always_inline lambda names a dependent type alias of the enclosing function -- the stub stays a
    // declaration and the library fails to link)
    constexpr int SLOTS = RrGeom<MODE>::SLOTS, ROW_F4 = RrGeom<MODE>::ROW_F4, SLOT_F4 = RrGeom<MODE>::SLOT_F4;
    constexpr int QUERY_WORDS = RrGeom<MODE>::QUERY_WORDS, AHEAD = RrGeom<MODE>::AHEAD, GT = RrGeom<MODE>::GT;
    constexpr int WAVES = RR_WAVES;
    constexpr int STAGE_ROWS = QT_ROWS / WAVES;      // rows of a query tile each wave stages: 4
    // ... in this many DMA instructions: one per row (64 lanes x 16 B = a 1 KiB row), or -- f16 x 1, 512 B per query -- one per PAIR
    // of rows.  The pair form needs the two rows contiguous in LDS (LDS-DMA writes lane-linear), i.e. NO row padding: bank
    // conflicts of the B reads are avoided by a swizzle instead -- the 16-B chunk c of row j sits at position c ^ (j & 15) of its
    // row, applied to the SOURCE address of the DMA and to the read (an involution on both sides, as in gemm_ldsrow_kernel).
    // Measured before (half-wave DMAs, one per row, padded rows): the staging instructions cost 0.73 of 5.9 ms at 1000 x 10 M;
    // two full-wave DMAs in their place: -0.37 ms.
    constexpr int STAGE_N = F16X1 ? STAGE_ROWS / 2 : STAGE_ROWS;
    constexpr int STAGE_EVERY = 2;                   // one DMA every so many K-steps at the start of a product
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4 *s_q = reinterpret_cast<f32x4 *>(smem_raw);                  // [4][32][65] float4: query tiles (split image)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    unsigned char *tbuf = smem_raw + SLOTS * SLOT_F4 * 16 + wave * RR_TBUF;
    unsigned char *s_qconst = smem_raw + SLOTS * SLOT_F4 * 16 + RR_WAVES * RR_TBUF;   // [slots][32] (threshold, 1/|q|)
    const uint32_t qs = blockIdx.x % p.qsplit;
    const uint32_t row_block = blockIdx.x / p.qsplit, row_blocks = gridDim.x / p.qsplit;
    const uint32_t qt_lo = (uint32_t)((uint64_t)qs * p.nqt / p.qsplit);
    const uint32_t qt_hi = (uint32_t)((uint64_t)(qs + 1) * p.nqt / p.qsplit);
    const uint32_t n_qt = qt_hi - qt_lo;
    const bool resident = n_qt <= (uint32_t)SLOTS;

    // row u of this wave's share of query tile qt -> LDS slot (padding rows of the image are zero rows: no branch)
    // f16 x 1: this lane's source offset (floats) inside pair u of the wave's four rows, swizzle included
    uint32_t pair_src[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t row_in_tile = (uint32_t)(wave * STAGE_ROWS + 2 * u + (lane >> 5));
        pair_src[u] = (uint32_t)(lane >> 5) * QUERY_WORDS + ((((uint32_t)lane & 31u) ^ (row_in_tile & 15u)) << 2);
    }
    uint32_t sw_low[8];   // f16 x 1: position of this lane's chunk of K-step m in its row, m mod 8
#pragma unroll
    for (int mm = 0; mm < 8; ++mm) sw_low[mm] = (uint32_t)((2 * mm + h) ^ (j & 15));
    auto stage_row = [&](uint32_t qt, int slot, int u) __attribute__((always_inline)) {
        const int r = wave * STAGE_ROWS + (F16X1 ? 2 * u : u);  // wave-uniform (f16 x 1: the first row of pair u)
        const uint32_t q = qt * QT_ROWS + r;
        f32x4 *dst = s_q + slot * SLOT_F4 + r * ROW_F4;
        if constexpr (F16X1) {   // rows r, r + 1: lane l carries chunk (l & 31) ^ (row & 15) of row r + (l >> 5) to position l & 31
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(p.queries_split) + (size_t)q * QUERY_WORDS + pair_src[u & 1],
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(p.queries_split) + (size_t)q * QUERY_WORDS + lane * 4,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    // the tile's 32 (threshold, 1/|q|) pairs: 256 B, one 4-byte DMA.  EVERY wave issues it (same bytes to the same
    // place) so that all waves count the same number of DMA instructions per tile -- the vmcnt arithmetic below.  (Measured
    // alternatives, 1000 x 10 M with the image: wave 0 alone issues it, with its own vmcnt count: 5.33 -> 5.58 ms -- the one wave
    // with more to do is the one the ring barrier waits for; the pairs of ALL tiles loaded once per launch into the 10 KiB of LDS
    // that are left: no better, 5.55 on a box that ran the unchanged 2000-query case 2 % slower.)
    auto stage_consts = [&](uint32_t qt, int slot) __attribute__((always_inline)) {
        __builtin_amdgcn_global_load_lds(p.qconst + (size_t)qt * QT_ROWS * 2 + lane,
                                         (__attribute__((address_space(3))) void *)(s_qconst + slot * RR_QCONST), 4, 0, 0);
    };
    auto stage_tile = [&](uint32_t qt, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < STAGE_N; ++u) stage_row(qt, slot, u);
        stage_consts(qt, slot);
    };
    auto stage_wait = [&]() { __builtin_amdgcn_s_waitcnt(0x0F70); };  // vmcnt(0)
    {
        const uint32_t first = resident ? n_qt : (uint32_t)((SMT_RR_EXP & 2) ? SLOTS : AHEAD);   // streaming: ring positions 0 .. AHEAD - 1
        for (uint32_t t = 0; t < first; ++t) stage_tile(qt_lo + t, (int)t);
        stage_wait();
    }
    __syncthreads();

    const uint64_t W = (uint64_t)row_blocks * WAVES;
    const uint64_t steps = (p.level_tiles - p.tile_begin + W - 1) / W;  // block-uniform trip count
    uint64_t it = p.tile_begin + (uint64_t)row_block * WAVES + wave;
    uint32_t pos = 0;            // streaming: running ring position (block-uniform); slot = pos & (SLOTS - 1)
    uint32_t tq = 0, tq_ahead = (uint32_t)AHEAD % n_qt;   // tile at position pos / pos + AHEAD (the tile sequence is cyclic over the sweeps)
    // transpose geometry: this lane WRITES row (8u + lane/8), bytes 8 * (lane%8) of a slice; it READS row j, quads 2mm + h
    const uint32_t t_wr = (uint32_t)((lane >> 3) * RR_TROW + (lane & 7) * 8);
    const uint32_t t_rd = (uint32_t)(j * RR_TROW + h * 16);

#if (SMT_RR_EXP & 8)
    u32x4 Ah[16], Al[F16X2 ? 1 : 16];
    unsigned zero16 = 0, valid16 = 0;
#endif
#if (SMT_RR_EXP & 256)
    int n_stamp = 0;
#endif
    uint32_t n_buf = 0;          // nominations waiting in this wave's LDS buffer (wave-uniform)
    for (uint64_t step = 0; step < steps; ++step, it += W) {
        const bool has = it < p.level_tiles;  // wave-uniform
        // (range-filtered: the level walks the tile table; the entry names the tile and which of its rows are wanted -- one scalar load)
        uint64_t tile_v = has ? level_tile(it, p.stride, p.skip16) : 0;
        uint32_t want32 = 0xffffffffu;
        if (p.tile_table != nullptr && has) {
            const uint64_t e = p.tile_table[tile_v];
            tile_v = e & 0xffffffffull;
            want32 = (uint32_t)(e >> 32);
        }
        const uint64_t row0 = tile_v * 32;
#if (SMT_RR_EXP & 256)
        const bool tracing = p.stamps != nullptr && blockIdx.x == 40 && step == 1 && p.level_tiles > 100000;
#endif
        RR_STAMP(1);   // step start
#if (SMT_RR_EXP & 256)
        // the shader clock under this load: s_memtime against the 100 MHz s_memrealtime, one step apart
        if (p.stamps != nullptr && blockIdx.x == 40 && wave == 0 && lane == 0 && (step == 1 || step == 9) && p.level_tiles > 100000) {
            p.stamps[8 * 1024 + (step == 1 ? 0 : 2)] = __builtin_readcyclecounter();
            p.stamps[8 * 1024 + (step == 1 ? 1 : 3)] = wall_clock64();
        }
#endif

#if !(SMT_RR_EXP & 8)
        u32x4 Ah[16], Al[F16X2 ? 1 : 16];
        unsigned zero16 = 0, valid16 = 0;
#endif
        if constexpr (IMG) {
            // ---- the operands are READY in the corpus' fp16 image (pack_image_kernel wrote them with the arithmetic of the branch
            // below): 16 loads of 1 KiB, quad m of lane l at 16 (64 m + l) in the tile's 16 KiB -- half the bytes of the f32 rows,
            // no norms, no conversion, no transpose.  (Requesting them one step ahead into a second register set was measured:
            // 8-32 queries 0.95 -> 0.94 ms, one query 0.92 -> 0.89, 256-512 queries 3-5 % slower with the spills it brings: not kept.)
            if (has) {
                const uint64_t tile = row0 >> 5;
                const u32x4 *img = reinterpret_cast<const u32x4 *>(p.image) + tile * 1024 + lane;
#pragma unroll
                for (int m = 0; m < 16; ++m) Ah[m] = __builtin_nontemporal_load(img + m * 64);
                RR_STAMP(2);
                if (n_buf) flush_candidates_lds(lane, tbuf, n_buf, p.cand, p.counts);
                const uint32_t zm = p.image_zero[tile] >> (4 * h);   // bit 8u + c: tile row 8u + c + 4h
                zero16 = 0;
                valid16 = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        zero16 |= ((zm >> (8 * u + c)) & 1u) << (4 * u + c);
                        if (row0 + 8 * u + c + 4 * h < p.n_rows && ((want32 >> (8 * u + c + 4 * h)) & 1u)) valid16 |= 1u << (4 * u + c);
                    }
            }
        } else
        if (has && (!(SMT_RR_EXP & 8) || step == 0)) {
            // ---- 32 coalesced loads: instruction i = 4s + u covers rows 8u .. 8u+7, dims 32s .. 32s+31 (128 B per row)
            f32x4 R[32];
            {
                // (rebuilt per tile from the SGPR base: as a loop invariant it was the one value hipcc spilled across the sweep)
                uint32_t l7 = (uint32_t)lane & 7u;
                asm volatile("" : "+v"(l7));
                const f32x4 *base = reinterpret_cast<const f32x4 *>(p.corpus) + l7;
                uint64_t rowv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint64_t r = row0 + 8 * u + (lane >> 3);
                    rowv[u] = r < p.n_rows ? r : 0;   // rows past the end: any valid address, masked by valid16
                }
#pragma unroll
                for (int i = 0; i < 32; ++i) R[i] = __builtin_nontemporal_load(base + rowv[i & 3] * 64 + 8 * (i >> 2));
            }
            RR_STAMP(2);   // row loads issued
            // the previous sweep's nominations leave the transpose buffer now: the atomics' round trip hides behind the row loads
            if (n_buf) flush_candidates_lds(lane, tbuf, n_buf, p.cand, p.counts);
            // ---- 1/|row| for the four rows this lane holds pieces of (8 lanes per row)
            float rb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float part = 0.0f;
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {
                    const f32x4 v = R[4 * sl + u];
                    part += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                }
                part += __shfl_xor(part, 1);
                part += __shfl_xor(part, 2);
                part += __shfl_xor(part, 4);
                rb[u] = part == 0.0f ? 0.0f : __frsqrt_rn(part);
            }
            RR_STAMP(3);   // rows arrived, norms done
            // ---- per slice: scale, split, transpose hi then lo through the wave's buffer
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                if constexpr (F16X2) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const f32x4 v = R[4 * sl + u] * (rb[u] * F16X2_ROW_SCALE);
                        *reinterpret_cast<u32x2 *>(tbuf + t_wr + u * 8 * RR_TROW) = (u32x2){f16_pack2(v.x, v.y), f16_pack2(v.z, v.w)};
                    }
                    Ah[2 * sl] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd);
                    Ah[2 * sl + 1] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd + 32);
                } else {
                    uint32_t hi[4][2], lo[4][2];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const f32x4 v = R[4 * sl + u] * rb[u];
                        bf16_split2(v.x, v.y, hi[u][0], lo[u][0]);
                        bf16_split2(v.z, v.w, hi[u][1], lo[u][1]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        *reinterpret_cast<u32x2 *>(tbuf + t_wr + u * 8 * RR_TROW) = (u32x2){hi[u][0], hi[u][1]};
                    Ah[2 * sl] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd);
                    Ah[2 * sl + 1] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd + 32);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        *reinterpret_cast<u32x2 *>(tbuf + t_wr + u * 8 * RR_TROW) = (u32x2){lo[u][0], lo[u][1]};
                    Al[2 * sl] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd);
                    Al[2 * sl + 1] = *reinterpret_cast<const u32x4 *>(tbuf + t_rd + 32);
                }
            }
            // accumulator reg r = 4u + c <-> tile row 8u + i, i = c + 4h; row 8u + i's scale sits in lanes 8i .. 8i + 7 of rb[u]:
            // one ballot per u instead of sixteen LDS permutes (whose hoisted lane indices were what spilled at 256 VGPRs)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned long long zm = __ballot(rb[u] == 0.0f);
                const uint32_t w = h ? (uint32_t)(zm >> 32) : (uint32_t)zm;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    zero16 |= ((w >> (8 * c)) & 1u) << (4 * u + c);
                    if (row0 + 8 * u + c + 4 * h < p.n_rows && ((want32 >> (8 * u + c + 4 * h)) & 1u)) valid16 |= 1u << (4 * u + c);
                }
            }
        }

        RR_STAMP(4);   // operands converted
        // K-steps M0 .. M1 - 1 of one (row tile x query tile) product.  While it runs the wave issues its share of the DMA
        // that brings query tile stage_qt into stage_slot (stage: wave-uniform): one row every second K-step, then the constants.
        auto product_part = [&](f32x16 &acc, int slot, const int M0, const int M1, bool stage, uint32_t stage_qt, int stage_slot) __attribute__((always_inline)) {
            // quad of (K-step m, half h): [hi, lo] pairs at 4 m + 2 h (+ 1) in the 1 KiB image; hi only at 2 m + h in the compact one
            constexpr int QS = F16X1 ? 2 : 4;   // quads per K-step in a query row
            const u32x4 *bq = reinterpret_cast<const u32x4 *>(s_q + slot * SLOT_F4 + j * ROW_F4) + (F16X1 ? 0 : 2 * h);
            const int M_CONSTS = M0 + (M1 - M0 > STAGE_EVERY * STAGE_N ? STAGE_EVERY * STAGE_N : M1 - M0 - 1);
            // f16 x 1: chunk 2m + h of row j sits at position (2m + h) ^ (j & 15): the low four bits are lane-dependent (eight values
            // per lane, m mod 8), the rest is the constant 16 (m >> 3).  (SIXTEEN lanes of a ds_read_b128 must fall into different
            // 16-byte bank groups: with c ^ (j & 7) the counters showed one conflict cycle per read cycle.)
            auto b_chunk = [&](int m) __attribute__((always_inline)) { return (int)sw_low[m & 7] + 16 * (m >> 3); };
            (void)b_chunk;
            if constexpr (F16X1) {
                // B quads arrive in GROUPS of four K-steps, double-buffered: wait for group g (an explicit lgkmcnt(0)), THEN
                // issue the four reads of group g + 1, THEN run the four MFMAs of group g -- the reads fly under 128 cycles of
                // this wave's MFMAs and no MFMA waits for a read issued an instruction earlier (hipcc guards a read issued
                // one or two K-steps ahead, as below for the other modes, with lgkmcnt(0) at every second MFMA).
                constexpr int BG = 4;   // (8, and 16 = no overlap inside a product at all, measure the same: 5.84-5.88 ms at 1000 x 10 M)
                u32x4 B[2][BG];
#pragma unroll
                for (int d = 0; d < BG; ++d) B[0][d] = bq[b_chunk(M0 + d)];
#pragma unroll
                for (int g = 0; g < (M1 - M0) / BG; ++g) {
                    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0), vmcnt / expcnt unconstrained: group g is in registers
                    if (g + 1 < (M1 - M0) / BG) {
#pragma unroll
                        for (int d = 0; d < BG; ++d) B[(g + 1) & 1][d] = bq[b_chunk(M0 + BG * (g + 1) + d)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int d = 0; d < BG; ++d) {
                        const int m = M0 + BG * g + d;
                        if ((m - M0) % STAGE_EVERY == 0 && (m - M0) / STAGE_EVERY < STAGE_N && stage) stage_row(stage_qt, stage_slot, (m - M0) / STAGE_EVERY);
                        if (m == M_CONSTS && stage) stage_consts(stage_qt, stage_slot);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Ah[m]), __builtin_bit_cast(f16x8, B[g & 1][d]), acc, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // B quads run RR_BDIST K-steps ahead of their MFMAs (sched_barrier: hipcc otherwise sinks every read to its
                // use and each K-step then starts with a full LDS round trip in front of 96 cycles of MFMA)
                // (bf16 x 3 holds 128 operand VGPRs: one K-step of distance keeps it at 256 registers WITHOUT spilling -- with two
                // it spilled three into scratch inside this loop; measured equal otherwise)
                constexpr int BD = MODE == 0 ? 1 : RR_BDIST;
                constexpr int NB = BD + 1;
                u32x4 bh[NB], bl[NB];
#pragma unroll
                for (int d = 0; d < BD; ++d) { bh[d] = bq[QS * (M0 + d)]; bl[d] = bq[QS * (M0 + d) + 1]; }
#pragma unroll
                for (int m = M0; m < M1; ++m) {
                    if (m + BD < M1) {
                        bh[(m - M0 + BD) % NB] = bq[QS * (m + BD)];
                        bl[(m - M0 + BD) % NB] = bq[QS * (m + BD) + 1];
                    }
                    if ((m - M0) % STAGE_EVERY == 0 && (m - M0) / STAGE_EVERY < STAGE_N && stage) stage_row(stage_qt, stage_slot, (m - M0) / STAGE_EVERY);
                    if (m == M_CONSTS && stage) stage_consts(stage_qt, stage_slot);
                    if constexpr (F16X2) acc = mfma_f16x2(Ah[m], bh[(m - M0) % NB], bl[(m - M0) % NB], acc);
                    else acc = mfma_bf16x3(Ah[m], Al[m], bh[(m - M0) % NB], bl[(m - M0) % NB], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        auto product_epilogue = [&](const f32x16 &acc, uint32_t qt, int slot) __attribute__((always_inline)) {
            // (thresholds from LDS, staged with the tile: a global load here would sit in the same in-order vmcnt queue as the
            // tile DMAs, and waiting for it would wait for them)
            const f32x2 qc = *reinterpret_cast<const f32x2 *>(s_qconst + slot * RR_QCONST + j * 8);
            if (p.tile_min != nullptr) {   // bootstrap level (wave-uniform): the tile's best nominating distance per query, a plain store
                float m;
                if (__builtin_amdgcn_ballot_w64(valid16 != 0xffffu)) {   // ragged / filtered tile: only the wanted rows count
                    m = -__builtin_inff();
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, (valid16 >> r) & 1u ? acc[r] : -__builtin_inff());
                } else {
                    m = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
                    for (int r = 3; r + 1 < 16; r += 2) m = fmaxf(fmaxf(m, acc[r]), acc[r + 1]);
                    m = fmaxf(m, acc[15]);
                }
                m = fmaxf(m, __shfl_xor(m, 32));                          // the other half's 16 rows of the same query
                const bool zero_row = __builtin_amdgcn_ballot_w64((zero16 & valid16) != 0u) != 0ull;   // simsimd: 0 against a zero row, else 1
                const float d = qc.y == 0.0f ? (zero_row ? 0.0f : 1.0f) : fmaxf(1.0f - m * qc.y, 0.0f);
                // slot = level tile index mod 1024: the tiles that share a slot are distinct (so are their best rows), the slot keeps
                // their minimum -- a fire-and-forget atomic (distances are >= 0: their bit patterns order like unsigned integers)
                if (h == 0)
                    (void)__hip_atomic_fetch_min(reinterpret_cast<unsigned int *>(p.tile_min) + (size_t)(it & 1023u) * ((size_t)p.nqt * QT_ROWS) +
                                                     (size_t)qt * QT_ROWS + j,
                                                 __float_as_uint(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
#if (SMT_RR_EXP & 1)
            if (acc[0] + acc[5] + acc[10] + acc[15] == 12345.678f)
#endif
            {
#if (SMT_RR_EXP & 256)
                if (p.buffered) append_candidates_lds(acc, zero16, valid16, qt * QT_ROWS + j, qc.x, qc.y, row0, h, lane, tbuf, n_buf, p.cand, p.counts,
                                                      tracing && n_stamp < 1000 ? p.stamps + wave * 1024 : nullptr, n_stamp);
#else
                if (p.buffered) append_candidates_lds(acc, zero16, valid16, qt * QT_ROWS + j, qc.x, qc.y, row0, h, lane, tbuf, n_buf, p.cand, p.counts);
#endif
                else append_candidates(acc, zero16, valid16, qt * QT_ROWS + j, qc.x, qc.y, row0, h, p.cand, p.counts);
            }
        };
        auto tile_product = [&](uint32_t qt, int slot, bool stage, uint32_t stage_qt, int stage_slot) __attribute__((always_inline)) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            RR_STAMP(5);   // product start
            product_part(acc, slot, 0, 16, stage, stage_qt, stage_slot);
            RR_STAMP(6);   // MFMAs issued
            product_epilogue(acc, qt, slot);
            RR_STAMP(7);   // epilogue done
        };

        if (resident) {
            if (has)
                for (uint32_t t = 0; t < n_qt; ++t) tile_product(qt_lo + t, (int)t, false, 0, 0);
        } else {
            for (uint32_t t = 0; t < n_qt; ++t) {
                const int slot = (int)(pos & (SLOTS - 1)), slot_ahead = (int)((pos + AHEAD) & (SLOTS - 1));   // slot_ahead was read during step pos - 1
#if (SMT_RR_EXP & 2)
                if (has) tile_product(qt_lo + tq, slot, false, qt_lo + tq_ahead, slot_ahead);
#else
                if (has) tile_product(qt_lo + tq, slot, true, qt_lo + tq_ahead, slot_ahead);
                else stage_tile(qt_lo + tq_ahead, slot_ahead);
#endif
                // this wave's share of the next group's tiles has landed: only the younger DMAs may still fly
                // (a raw s_barrier: __syncthreads() carries a fence that hipcc lowers to vmcnt(0), i.e. it would wait
                // for the tiles that are meant to stay in flight.  The LDS reads of this step were consumed by MFMAs.)
                // At the border in front of position pos + 1 the tiles pos + 1 .. pos + GT must have landed; the younger ones,
                // pos + GT + 1 .. pos + AHEAD, may still fly.
                constexpr int FLY = (AHEAD - GT) * (STAGE_N + 1);                  // 10 (four slots, GT 1) / 12 (eight, GT 2, pairs)
                static_assert(FLY >= 0 && FLY < 64, "vmcnt is a 6-bit counter");
                if ((pos + 1) % GT == 0 && !(SMT_RR_EXP & 4)) {   // block-uniform
                    __builtin_amdgcn_s_waitcnt(0x0F70 | (FLY & 15) | ((FLY >> 4) << 14));  // vmcnt(FLY), expcnt / lgkmcnt unconstrained
                    RR_STAMP(8);   // own DMAs landed
                    __builtin_amdgcn_s_barrier();
                    RR_STAMP(9);   // barrier passed
                }
                asm volatile("" ::: "memory");
                ++pos;
                tq = tq + 1 == n_qt ? 0 : tq + 1;
                tq_ahead = tq_ahead + 1 == n_qt ? 0 : tq_ahead + 1;
            }
        }
    }
    if (n_buf) flush_candidates_lds(lane, tbuf, n_buf, p.cand, p.counts);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): no LDS-DMA may outlive the block's LDS allocation
}

hipError_t gemm_rowreg_set_attrs()
{
    const void *kernels[] = {reinterpret_cast<const void *>(gemm_rowreg_kernel<0>), reinterpret_cast<const void *>(gemm_rowreg_kernel<1>),
                             reinterpret_cast<const void *>(gemm_rowreg_kernel<2>), reinterpret_cast<const void *>(gemm_rowreg_kernel<1, true>),
                             reinterpret_cast<const void *>(gemm_rowreg_kernel<2, true>)};
    for (const void *k : kernels) {
        const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

void gemm_rowreg_launch(smt_ctx *ctx, int mode, bool image, int nb, const GemmParams &g)
{
    if (mode == 2 && image) hipLaunchKernelGGL((gemm_rowreg_kernel<2, true>), dim3(nb), dim3(RR_THREADS), (size_t)RrGeom<2>::SMEM, ctx->stream, g);
    else if (mode == 1 && image) hipLaunchKernelGGL((gemm_rowreg_kernel<1, true>), dim3(nb), dim3(RR_THREADS), (size_t)RrGeom<1>::SMEM, ctx->stream, g);
    else if (mode == 2) hipLaunchKernelGGL(gemm_rowreg_kernel<2>, dim3(nb), dim3(RR_THREADS), (size_t)RrGeom<2>::SMEM, ctx->stream, g);
    else if (mode == 1) hipLaunchKernelGGL(gemm_rowreg_kernel<1>, dim3(nb), dim3(RR_THREADS), (size_t)RrGeom<1>::SMEM, ctx->stream, g);
    else hipLaunchKernelGGL(gemm_rowreg_kernel<0>, dim3(nb), dim3(RR_THREADS), (size_t)RrGeom<0>::SMEM, ctx->stream, g);
}

int launch_pack_image(smt_ctx *ctx, const float *corpus, uint64_t n_rows, uint64_t first_tile, uint64_t n_tiles, void *image,
                      uint32_t *image_zero)
{
    if (n_tiles == 0) return SMT_OK;
    prof_begin(ctx, "pack_image");
    hipLaunchKernelGGL(pack_image_kernel, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, ctx->stream, corpus, n_rows, first_tile,
                       n_tiles, reinterpret_cast<uint32_t *>(image), image_zero);
    prof_end(ctx, "pack_image");
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

}  // namespace smt
