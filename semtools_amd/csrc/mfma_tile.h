// mfma_tile.h -- constants/types shared by the f32-MFMA kernels (K3 batched search, IVF assignment).
// Geometry: a wave keeps a 32-row A tile in 128 VGPRs; B tiles of 32 vectors x 256 dims stream
// through LDS (1040-B row stride => conflict-free ds_read_b128); 128 v_mfma_f32_32x32x2_f32 per
// (A tile, B tile); lane l < 32 feeds dims 8m..8m+3, lane l >= 32 dims 8m+4..8m+7 of group m.
#pragma once
#include "common.h"
#include "device_utils.h"

namespace smt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_THREADS = 512;              // 8 waves: 2 per SIMD
constexpr int GEMM_WAVES = GEMM_THREADS / 64;
constexpr int QT_ROWS = 32;                    // B vectors per tile
constexpr int QT_STRIDE_F4 = 65;               // 1040-B LDS rows: conflict-free ds_read_b128
constexpr int QT_F4 = QT_ROWS * QT_STRIDE_F4;  // float4 per staged tile

// acc += A(32 rows in registers) x B(tile in LDS)^T for this lane's column j / half h
__device__ __forceinline__ f32x16 mfma_tile_32x32x256(const f32x4 (&A)[32], const f32x4 *bq)
{
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        const f32x4 b = bq[2 * m];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m].w, b.w, acc, 0, 0, 0);
    }
    return acc;
}

// ---- bf16 x 3 split products on the bf16 MFMA pipe (v_mfma_f32_32x32x16_bf16: 16x the f32 MFMA rate).
// An f32 value x is split into hi = bf16(x) (round to nearest even: 8 significant bits, |x - hi| <= 2^-8 |x|) and
// lo = bf16(x - hi): x - hi is exact in f32, so |x - hi - lo| <= 2^-16 |x|.  x . q  ~=  xh.qh + xl.qh + xh.ql  (three
// MFMAs into ONE f32 accumulator; the products of bf16 pairs are exact in f32).  Dropped: xl.ql (<= 2^-8 * 2^-8) and
// the two residual terms, each <= 2^-16 sum |x_i q_i| <= 2^-16 |x||q| (Cauchy-Schwarz) => <= 3 * 2^-16 |x||q| =
// 4.6e-5 |x||q|; the accumulation was MEASURED (tools/micro/mfma_rounding.hip, profiles/r03_mfma_rounding.json): a 16-bit MFMA
// forms its 16 products exactly and rounds their sum into the accumulator ONCE, <= 2 ulp per instruction -- 48 instructions
// add <= 1.1e-5.  common.h: F32_ERR_BF16X3 = 7e-5 (5.8e-5 derived + margin; the rounding property is re-checked on the device by
// tests/test_gpu_batched.py).  Constructed worst case: 2.6e-5; random corpora: 1.0e-5.  The scores only
// NOMINATE candidates; final distances are recomputed exactly (f64) and the certificate of section 5 of DESIGN.md uses
// this bound.
// Operand layout: lane (j, h) feeds row / query j with dims 16m + 8h .. + 7 of K-step m (8 bf16 = 4 VGPRs), the same
// K permutation on both operands.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t bf16_pack2(float a, float b)  // v_cvt_pk_bf16_f32: a -> bits 15:0, b -> bits 31:16
{
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void bf16_split2(float a, float b, uint32_t &hi, uint32_t &lo)
{
    hi = bf16_pack2(a, b);
    const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xFFFF0000u);
    lo = bf16_pack2(a - ah, b - bh);
}
// eight consecutive dims (two float4) -> hi and lo operand quads
__device__ __forceinline__ void bf16_split8(const f32x4 &a, const f32x4 &b, u32x4 &hi, u32x4 &lo)
{
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
    bf16_split2(a.x, a.y, h0, l0);
    bf16_split2(a.z, a.w, h1, l1);
    bf16_split2(b.x, b.y, h2, l2);
    bf16_split2(b.z, b.w, h3, l3);
    hi = (u32x4){h0, h1, h2, h3};
    lo = (u32x4){l0, l1, l2, l3};
}
// acc += (ah + al) . (bh + bl) without the al.bl term
__device__ __forceinline__ f32x16 mfma_bf16x3(const u32x4 &ah, const u32x4 &al, const u32x4 &bh, const u32x4 &bl, f32x16 acc)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), acc, 0, 0, 0);
    return acc;
}

// ---- f16 x 2: one fp16 operand for the rows, hi + lo fp16 for the queries -- TWO MFMAs per 16 dims instead of three.
// fp16 keeps 11 significant bits: a row element rounds with relative error <= 2^-11, so the score is off by at most
// 2^-11 |x||q| = 4.9e-4 (the query, hi + lo, carries 22 bits: its residual is noise next to that); 32 instructions x 2 ulp
// (measured, see above) add <= 7.6e-6.  common.h: F32_ERR_F16X2 = 5.2e-4, seven times the bf16 x 3 bound -- the price of a
// third fewer MFMAs, which is what the part's power budget is spent on in large batches.  Scaling keeps everything away
// from fp16's subnormals: unit rows are multiplied by 2^10, unit queries by 2^8 (an element would have to be below
// 6e-8 resp. 2.4e-7 to be flushed); the accumulator then holds 2^18 cos, which the query's constant 1/|q| := 2^-18 undoes.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float F16X2_ROW_SCALE = 1024.0f, F16X2_QUERY_SCALE = 256.0f, F16X2_INV_SCALE = 1.0f / (1024.0f * 256.0f);

__device__ __forceinline__ uint32_t f16_pack2(float a, float b)  // v_cvt_pk_f16_f32 (round to nearest even)
{
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ void f16_split2(float a, float b, uint32_t &hi, uint32_t &lo)
{
    const f32x2 v = {a, b};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = f16_pack2(a - hf.x, b - hf.y);
}
// acc += a . (bh + bl)
__device__ __forceinline__ f32x16 mfma_f16x2(const u32x4 &a, const u32x4 &bh, const u32x4 &bl, f32x16 acc)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bl), acc, 0, 0, 0);
    return acc;
}

// accumulator register r of lane (j, h) holds tile row acc_row(r, h) and tile column j
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

}  // namespace smt
