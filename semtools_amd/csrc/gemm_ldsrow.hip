// gemm_ldsrow.hip -- gemm_ldsrow_kernel: the K3 kernel of range-filtered batches (and of small batches when gemm_rowreg = 0).
// DESIGN.md 4.3, profiles/HISTORY.md 4.3b.
#include "gemm.h"

namespace smt {

// ---- small / medium batches on f32 MFMAs and every range-filtered batch (nq <= 64 per pass): ROW TILES LAND IN LDS
// BY LDS-DMA.  (The first-generation kernel of this regime, removed since, pulled its 32-row tile straight into
// registers with fragment-shaped loads: one wave instruction touches 32 B of 32 different rows -- 32 queries x 10 M
// rows took 3.0 ms against an HBM bound of 1.28 ms.  gemm_rowreg_kernel above is the third answer to the same
// problem and the default for unfiltered batches.)  Here
//   * the corpus streams through a per-wave LDS ring of 8 KiB K-SLICES (32 rows x 64 dims) filled by
//     global_load_lds_dwordx4: one instruction moves 4 rows x 256 contiguous bytes, no staging registers, no ds_write;
//   * A fragments are read from the ring with ds_read_b128 (one read feeds 4 x NQT MFMAs); a slot is refilled the
//     moment its fragments sit in registers, so the ring only ever holds bytes in flight;
//   * NQT = 1 (<= 32 queries, HBM-bound): the query tile is the register-resident MFMA operand (128 VGPRs), the
//     ring has two slots per wave (16 KiB/wave, 128 KiB/CU in flight), counted s_waitcnt vmcnt(8);
//     NQT = 2 (<= 64 queries, MFMA-bound): both operands come from LDS (64 KiB of swizzled queries + one slot per
//     wave), the 64 MFMAs of a slice cover the refill of its slot;
//   * EIGHT waves per CU, two per SIMD (<= 256 VGPRs): the first version ran one wave per SIMD and measured
//     MFMA time + everything else, back to back (ablations on MI355X, 32 queries x 10 M rows: MFMAs alone 1.04 ms,
//     LDS reads + address work alone 0.63 ms, together 1.68 ms; epilogue +0.25 ms; DMA waits +0.25 ms) -- a lone
//     wave issues in order, so its own LDS waits, norm FMAs and epilogue stall its MFMA stream; the second wave of
//     the SIMD fills those holes (gemm_level_kernel already worked that way);
//   * no barrier anywhere: every wave runs its own pipeline (s_waitcnt vmcnt(N) covers the issuing wave's LDS-DMA).
// LDS image: rows are 256 B apart inside a slice (no padding: LDS-DMA writes lane-linear), so the 16-B chunk c of
// row i is stored at position c ^ (i & 7): the swizzle is applied to the SOURCE address of the DMA and to the
// fragment read (same involution on both sides), which spreads the 8 lanes of a read phase over all 32 banks.
// Row norms come from the fragments (each lane squares the half row it reads anyway); the scale 1/|row| moves to
// the epilogue (acc * rb >= threshold) and travels through a 128-B LDS scratch (one write, four b128 reads per lane
// instead of 16 ds_bpermute round trips).
// FILTERED: tile t = chunks 8t .. 8t+7 of the chunk table (4 rows each, the last of a range short): DMA instruction
// u of a slice covers exactly chunk u, whose descriptor is a wave-uniform scalar load.

// LDS-DMA of one K-slice (dims 64*S ..) of a tile into `slot`: 8 instructions, each moves 4 rows x 256 B (1 KiB,
// lane-linear in LDS).  rows[u] = this lane's row for instruction u; swz = its swizzled chunk offset in floats for
// even / odd u.  (A free function template, not a generic lambda inside the kernel: with
// __builtin_amdgcn_global_load_lds inside a generic lambda hipcc 7.2 silently drops the kernel's HOST stub.)
template <int S, int AUX>
__device__ __forceinline__ void lr_fill_slice(const float *corpus, const uint32_t (&rows)[8], uint32_t swz_even, uint32_t swz_odd,
                                              unsigned char *slot)
{
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const float *src = corpus + (uint64_t)rows[u] * 256 + ((u & 1) ? swz_odd : swz_even) + 64 * S;
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(slot + u * 1024), 16, 0, AUX);
    }
}

// AUX: cache-policy bits of the row DMA (0 = default, 2 = nt: the corpus is streamed once)
// BF16: bf16 x 3 split products (mfma_tile.h).  The rows still arrive as f32 (LDS-DMA moves bytes); a wave splits the
// fragments it reads (2.5 VALU instructions per element, next to 3 MFMAs of 32 cycles per 8 elements instead of 8 MFMAs
// of 64) -- the MFMA pipe drops from 55 % busy to 10 % at 32 queries and the kernel is purely a question of row arrival.
template <int NQT, bool FILTERED, int AUX, bool BF16>
__global__ void __launch_bounds__(LR_THREADS) gemm_ldsrow_kernel(GemmParams p)
{
    using G = LrGeom<NQT>;
    constexpr bool B_REGS = NQT == 1;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    unsigned char *ring = smem_raw + G::Q_BYTES + wave * G::RING_BYTES;
    float *scale = reinterpret_cast<float *>(smem_raw + G::SCRATCH_OFF + wave * 128);

    // fragment geometry: lane (j, h) reads chunk 2m'+h of row j, stored at position chunk ^ (j & 7)
    // (BF16: K-step m'' of the slice takes chunks 4m'' + 2h and + 1: fragments 2m'', 2m'' + 1)
    uint32_t foff[8];  // byte offset of fragment m' inside a slice
#pragma unroll
    for (int mp = 0; mp < 8; ++mp) {
        const int chunk = BF16 ? 4 * (mp >> 1) + 2 * h + (mp & 1) : 2 * mp + h;
        foff[mp] = (uint32_t)(j * 256 + ((chunk ^ (j & 7)) << 4));
    }

    // ---- B operand: this lane's query of every tile, K-permuted like the A fragments (dims 8m+4h .. +3 in group m)
    f32x4 Bq[B_REGS && !BF16 ? 32 : 1];
    u32x4 Bh[B_REGS && BF16 ? 16 : 1], Bl[B_REGS && BF16 ? 16 : 1];
    float thr[NQT], rq[NQT];
    if constexpr (!B_REGS) {
        // queries -> LDS by LDS-DMA, one 1 KiB row per instruction, chunk c of row r at position c ^ (r & 7)
        // (BF16: the rows of the split image, see split_queries_kernel)
        const float *qsrc = BF16 ? reinterpret_cast<const float *>(p.queries_split) : p.queries;
        for (int r = wave; r < NQT * QT_ROWS; r += LR_WAVES) {  // wave-uniform
            unsigned char *dst = smem_raw + r * 1024;
            if ((uint32_t)r < p.nq)
                __builtin_amdgcn_global_load_lds(qsrc + (size_t)r * 256 + ((lane ^ (r & 7)) << 2),
                                                 (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            else
                reinterpret_cast<f32x4 *>(dst)[lane] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __syncthreads();                     // the only barrier: the query image is shared by the block's waves
    }
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        const uint32_t q = t * QT_ROWS + j;
        const bool ok = q < p.nq;
        float part = 0.0f;
        if constexpr (BF16) {
            // the norm always comes from the f32 query; with B in registers the lane also splits its operand quads
            const f32x4 *src = reinterpret_cast<const f32x4 *>(p.queries + (size_t)(ok ? q : 0) * 256) + 2 * h;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                f32x4 v0 = src[4 * m], v1 = src[4 * m + 1];
                if (!ok) v0 = v1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                part += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
                part += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
                if constexpr (B_REGS) bf16_split8(v0, v1, Bh[m], Bl[m]);
            }
        } else if constexpr (B_REGS) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(p.queries + (size_t)(ok ? q : 0) * 256) + h;
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                f32x4 v = src[2 * m];
                if (!ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                Bq[m] = v;
                part += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        } else {
            const unsigned char *qrow = smem_raw + t * LR_QTILE_BYTES + j * 1024;
#pragma unroll 8
            for (int m = 0; m < 32; ++m) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(qrow + (((2 * m + h) ^ (j & 7)) << 4));
                part += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        }
        const float a2 = part + __shfl_xor(part, 32);
        rq[t] = a2 == 0.0f ? 0.0f : __frsqrt_rn(a2);
        thr[t] = score_threshold(ok ? p.tau[q] : -1.0f, rq[t]);  // padding: zero query, tau < 0 -> never passes
    }
    // (every ordinary global load above has been consumed: none is pending when the first row DMA is issued)

    const uint64_t W = (uint64_t)gridDim.x * LR_WAVES;
    uint64_t it = (uint64_t)blockIdx.x * LR_WAVES + wave;
    if (it >= p.level_tiles) return;
    const const_u64_ptr table = (const_u64_ptr)(uintptr_t)p.chunk_table;

    // lane geometry of one DMA instruction: rows 4u .. 4u+3 of the tile, 256 B of each
    const int rl = lane >> 4, pos = lane & 15;
    const uint32_t swz_even = (uint32_t)((pos ^ rl) << 2);        // tile row i = 4u + rl: i & 7 = rl (u even)
    const uint32_t swz_odd = (uint32_t)((pos ^ (4 + rl)) << 2);   //                             4 + rl (u odd)

    struct TileSrc {
        uint32_t rows[8];      // per DMA instruction: this lane's corpus row
        uint32_t row0[8];      // FILTERED: first row of chunk u (wave-uniform)
        uint32_t valid32;      // bit i: tile row i exists (wave-uniform)
        uint32_t first_row;    // unfiltered: row of tile row 0
    };
    auto describe = [&](uint64_t tile, TileSrc &d) {
        d.valid32 = 0;
        d.first_row = (uint32_t)(tile * 32);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (FILTERED) {
                const uint64_t c = tile * 8 + u;
                const uint64_t desc = c < p.n_chunks ? table[c] : 0ull;  // wave-uniform: scalar load
                const uint32_t r0 = (uint32_t)desc, cnt = (uint32_t)(desc >> 32);
                d.row0[u] = r0;
                d.valid32 |= ((1u << cnt) - 1u) << (4 * u);
                d.rows[u] = r0 + ((uint32_t)rl < cnt ? (uint32_t)rl : (cnt ? cnt - 1 : 0u));
            } else {
                d.row0[u] = 0;
                const uint64_t row = tile * 32 + 4 * u + rl;
                d.rows[u] = (uint32_t)(row < p.n_rows ? row : p.n_rows - 1);  // clamp: fetched, never used (valid32)
            }
        }
        if constexpr (!FILTERED) {
            const uint64_t left = p.n_rows > tile * 32 ? p.n_rows - tile * 32 : 0;
            d.valid32 = left >= 32 ? 0xFFFFFFFFu : ((1u << (uint32_t)left) - 1u);
        }
    };
    auto fill = [&](const TileSrc &d, auto S, int slot) {
        lr_fill_slice<decltype(S)::value, AUX>(p.corpus, d.rows, swz_even, swz_odd, ring + slot * LR_SLICE_BYTES);
    };
    auto read_frags = [&](int slot, f32x4 (&f)[8]) {
#pragma unroll
        for (int mp = 0; mp < 8; ++mp) f[mp] = *reinterpret_cast<const f32x4 *>(ring + slot * LR_SLICE_BYTES + foff[mp]);
    };
    // counted waits (imm: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14)
    auto wait_landed = [&]() {   // the NEXT slice has landed; with two slots 8 younger instructions may still fly
        if constexpr (G::SLOTS == 2) __builtin_amdgcn_s_waitcnt(0x0F78);  // vmcnt(8)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0)
        asm volatile("" ::: "memory");
    };
    auto wait_lgkm0 = [&]() { __builtin_amdgcn_s_waitcnt(0xC07F); asm volatile("" ::: "memory"); };  // lgkmcnt(0), vmcnt untouched

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    TileSrc cur_t, nxt_t;
    describe(level_tile(it, p.stride, p.skip16), cur_t);
    fill(cur_t, S0{}, 0);
    if constexpr (G::SLOTS == 2) fill(cur_t, S1{}, 1);
    wait_landed();
    f32x4 fr[8];
    read_frags(0, fr);

    for (; it < p.level_tiles; it += W) {
        const bool more = it + W < p.level_tiles;  // wave-uniform
        if (more) describe(level_tile(it + W, p.stride, p.skip16), nxt_t);
        else nxt_t = cur_t;                        // dummy refills keep the vmcnt arithmetic uniform (valid addresses)

        f32x16 acc[NQT];
#pragma unroll
        for (int t = 0; t < NQT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        float part = 0.0f;

        // One K-slice.  `fr` holds slice S's fragments (its slot is free): refill that slot with the slice SLOTS
        // ahead, multiply, then pull the next slice's fragments into `fr`.
        auto step = [&](auto S, auto SAHEAD, const TileSrc &ahead_tile) {
            constexpr int s = decltype(S)::value;
            constexpr int slot = G::SLOTS == 2 ? (s & 1) : 0;
            constexpr int next_slot = G::SLOTS == 2 ? ((s + 1) & 1) : 0;
            wait_lgkm0();                              // the ds_reads of `fr` have returned: its slot is free
            fill(ahead_tile, SAHEAD, slot);
            if constexpr (BF16) {
#pragma unroll
                for (int mq = 0; mq < 4; ++mq) {       // K-step 4s + mq
                    const f32x4 a0 = fr[2 * mq], a1 = fr[2 * mq + 1];
                    part += a0.x * a0.x + a0.y * a0.y + a0.z * a0.z + a0.w * a0.w;
                    part += a1.x * a1.x + a1.y * a1.y + a1.z * a1.z + a1.w * a1.w;
                    u32x4 ah, al;
                    bf16_split8(a0, a1, ah, al);
#pragma unroll
                    for (int t = 0; t < NQT; ++t) {
                        u32x4 bh, bl;
                        if constexpr (B_REGS) {
                            bh = Bh[4 * s + mq];
                            bl = Bl[4 * s + mq];
                        } else {
                            const unsigned char *qrow = smem_raw + t * LR_QTILE_BYTES + j * 1024;
                            const int c = 2 * (2 * (4 * s + mq) + h);  // chunk of the hi quad; lo is the next one
                            bh = *reinterpret_cast<const u32x4 *>(qrow + ((c ^ (j & 7)) << 4));
                            bl = *reinterpret_cast<const u32x4 *>(qrow + (((c + 1) ^ (j & 7)) << 4));
                        }
                        acc[t] = mfma_bf16x3(ah, al, bh, bl, acc[t]);
                    }
                }
            } else
#pragma unroll
            for (int mp = 0; mp < 8; ++mp) {
                const f32x4 a = fr[mp];
                part += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
#pragma unroll
                for (int t = 0; t < NQT; ++t) {
                    f32x4 b;
                    if constexpr (B_REGS) b = Bq[8 * s + mp];
                    else b = *reinterpret_cast<const f32x4 *>(smem_raw + t * LR_QTILE_BYTES + j * 1024 +
                                                              (((2 * (8 * s + mp) + h) ^ (j & 7)) << 4));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);         // the MFMAs stay above: `fr` is single-buffered (register budget)
            wait_landed();                             // the next slice is in LDS
            read_frags(next_slot, fr);
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (G::SLOTS == 2) {
            step(S0{}, S2{}, cur_t);   // slot 0 <- (this tile, slice 2)
            step(S1{}, S3{}, cur_t);   // slot 1 <- (this tile, slice 3)
            step(S2{}, S0{}, nxt_t);   // slot 0 <- (next tile, slice 0)
            step(S3{}, S1{}, nxt_t);   // slot 1 <- (next tile, slice 1); leaves the next tile's slice-0 fragments in fr
        } else {
            step(S0{}, S1{}, cur_t);
            step(S1{}, S2{}, cur_t);
            step(S2{}, S3{}, cur_t);
            step(S3{}, S0{}, nxt_t);
        }

        // ---- epilogue: lane (j, h) owns query j of every tile and the 16 rows acc_row(r, h)
        const float b2 = part + __shfl_xor(part, 32);   // row j's norm^2 (both halves hold it)
        const float rb = b2 == 0.0f ? 0.0f : __frsqrt_rn(b2);
        if (h == 0) scale[j] = rb;                      // wave-private scratch: no barrier, an lgkmcnt wait orders it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float rbv[16];
        unsigned zero16 = 0;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + 8 * gq + 4 * h);  // rows acc_row(4gq .. 4gq+3, h)
            rbv[4 * gq + 0] = sc.x; rbv[4 * gq + 1] = sc.y; rbv[4 * gq + 2] = sc.z; rbv[4 * gq + 3] = sc.w;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (rbv[r] == 0.0f) zero16 |= 1u << r;
        const uint32_t v = cur_t.valid32 >> (4 * h);
        const unsigned valid16 = (v & 0xFu) | (((v >> 8) & 0xFu) << 4) | (((v >> 16) & 0xFu) << 8) | (((v >> 24) & 0xFu) << 12);
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            const uint32_t q = t * QT_ROWS + j;
            unsigned pass = 0;
            auto dist_of = [&](int r) {
                if (rq[t] == 0.0f) return (zero16 >> r) & 1u ? 0.0f : 1.0f;   // zero query (simsimd rules)
                return fmaxf(1.0f - acc[t][r] * rbv[r] * rq[t], 0.0f);         // a zero row has rb == 0 -> 1
            };
            if (rq[t] != 0.0f) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (acc[t][r] * rbv[r] >= thr[t]) pass |= 1u << r;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (dist_of(r) <= thr[t]) pass |= 1u << r;
            }
            pass &= valid16;
            if (__builtin_amdgcn_ballot_w64(pass != 0)) {
                if (pass) {
                    const unsigned base = atomicAdd(&p.counts[q], (unsigned)__popc(pass));
                    key_t64 *dst = p.cand + (size_t)q * CAND_CAP;
                    unsigned slot = base;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (pass & (1u << r)) {
                            const int i = acc_row(r, h);
                            // (constant indices only: a lane-dependent index would send row0[] to scratch)
                            const uint32_t r0 = h ? cur_t.row0[2 * (r >> 2) + 1] : cur_t.row0[2 * (r >> 2)];
                            const uint32_t row = FILTERED ? r0 + (uint32_t)(i & 3) : cur_t.first_row + (uint32_t)i;
                            if (slot < CAND_CAP) dst[slot] = make_key(dist_of(r), row);
                            ++slot;
                        }
                    }
                }
            }
        }
        cur_t = nxt_t;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): no LDS-DMA may outlive the wave's LDS allocation
}

template <int NQT, bool FILTERED, int AUX, bool BF16>
static hipError_t lr_attr()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_ldsrow_kernel<NQT, FILTERED, AUX, BF16>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <bool BF16>
static hipError_t lr_set_attr()
{
    hipError_t e;
    if ((e = lr_attr<1, false, 0, BF16>()) != hipSuccess) return e;
    if ((e = lr_attr<2, false, 0, BF16>()) != hipSuccess) return e;
    if ((e = lr_attr<1, false, 2, BF16>()) != hipSuccess) return e;
    if ((e = lr_attr<2, false, 2, BF16>()) != hipSuccess) return e;
    if ((e = lr_attr<1, true, 0, BF16>()) != hipSuccess) return e;
    return lr_attr<2, true, 0, BF16>();
}
template <int NQT, bool FILTERED, int AUX, bool BF16>
static void lr_launch(smt_ctx *ctx, int nb, size_t smem, const GemmParams &g)
{
    hipLaunchKernelGGL((gemm_ldsrow_kernel<NQT, FILTERED, AUX, BF16>), dim3(nb), dim3(LR_THREADS), smem, ctx->stream, g);
}

hipError_t gemm_ldsrow_set_attrs()
{
    hipError_t e = lr_set_attr<false>();
    if (e != hipSuccess) return e;
    return lr_set_attr<true>();
}

template <bool BF16>
static void lr_dispatch(smt_ctx *ctx, uint32_t nqt, bool filtered, bool nt, int nb, size_t smem, const GemmParams &g)
{
    if (nqt <= 1) {
        if (filtered) lr_launch<1, true, 0, BF16>(ctx, nb, smem, g);
        else if (nt) lr_launch<1, false, 2, BF16>(ctx, nb, smem, g);
        else lr_launch<1, false, 0, BF16>(ctx, nb, smem, g);
    } else {
        if (filtered) lr_launch<2, true, 0, BF16>(ctx, nb, smem, g);
        else if (nt) lr_launch<2, false, 2, BF16>(ctx, nb, smem, g);
        else lr_launch<2, false, 0, BF16>(ctx, nb, smem, g);
    }
}

void gemm_ldsrow_launch(smt_ctx *ctx, bool bf16, uint32_t nqt, bool filtered, bool nt, int nb, const GemmParams &g)
{
    const size_t smem = nqt <= 1 ? (size_t)LrGeom<1>::SMEM : (size_t)LrGeom<2>::SMEM;
    if (bf16) lr_dispatch<true>(ctx, nqt, filtered, nt, nb, smem, g);
    else lr_dispatch<false>(ctx, nqt, filtered, nt, nb, smem, g);
}

}  // namespace smt
