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

// accumulator register r of lane (j, h) holds tile row acc_row(r, h) and tile column j
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

}  // namespace smt
