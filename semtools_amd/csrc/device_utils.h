// device_utils.h -- wave64 / DPP helpers shared by the gfx950 kernels.
#pragma once
#include "common.h"

namespace smt {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ DPP helpers
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ float readlane_f(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i within 8
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15-i within 16
constexpr int DPP_WAVE_SHR1 = 0x138;   // lane i <- lane i-1 across the wave

// Sum over the 64 lanes, result in every lane.  Fixed tree => deterministic.
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    v += dpp_f<DPP_MIRROR>(v);
    const float s0 = readlane_f(v, 0), s1 = readlane_f(v, 16);
    const float s2 = readlane_f(v, 32), s3 = readlane_f(v, 48);
    return (s0 + s1) + (s2 + s3);
}

// FOUR sums over the 64 lanes at once: lane l of the result holds the sum over all lanes of the input with index l % 4 (a, b, c, d).
// A transposing tree: the xor-1 and xor-2 steps halve the number of live values instead of repeating each step per value, rotations
// by 4 and 8 finish the 16-lane rows, v_permlane16_swap / v_permlane32_swap (gfx950) add the rows -- 15 VALU instructions for four
// sums against 4 x 11 of wave_sum (whose four v_readlane + scalar adds per value are most of its cost).  Fixed tree => deterministic.
constexpr int DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128;   // lane i of a row <- lane (i - n) mod 16
// (v_permlane{16,32}_swap_b32 through inline asm: __builtin_amdgcn_permlane*_swap of clang 22 / ROCm 7.2 hands back its FIRST result
// for both elements -- the generated code added v + v -- so the builtin cannot give "lane + partner".  The two wait states a VALU
// write needs before the swap reads the register are in the asm, the assembler adds none for inline code.
// tools/micro/wave_sum4.hip is the on-device test of what comes out.)
__device__ __forceinline__ float add_lane_xor32(float v)
{
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1" : "+v"(a), "+v"(b));   // a = {lo, lo}, b = {hi, hi}
    return a + b;
}
__device__ __forceinline__ float add_lane_xor16(float v)
{
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32_e32 %0, %1" : "+v"(a), "+v"(b));   // a = rows {0, 0, 2, 2}, b = rows {1, 1, 3, 3}
    return a + b;
}
__device__ __forceinline__ float wave_sum4(float a, float b, float c, float d, int lane)
{
    const bool odd = lane & 1, upper = lane & 2;
    // xor 1: even lanes keep a (c), odd lanes keep b (d), each adds its neighbour's share of the value it keeps
    const float ab = (odd ? b : a) + dpp_f<DPP_XOR1>(odd ? a : b);
    const float cd = (odd ? d : c) + dpp_f<DPP_XOR1>(odd ? c : d);
    // xor 2: lanes 0, 1 of a quad keep a | b, lanes 2, 3 keep c | d
    float v = (upper ? cd : ab) + dpp_f<DPP_XOR2>(upper ? ab : cd);
    v += dpp_f<DPP_ROW_ROR4>(v);
    v += dpp_f<DPP_ROW_ROR8>(v);
    v = add_lane_xor16(v);
    return add_lane_xor32(v);
}

// Integer sum over the 64 lanes, result in every lane.
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    v += dpp_u<DPP_XOR1>(v);
    v += dpp_u<DPP_XOR2>(v);
    v += dpp_u<DPP_HALF_MIRROR>(v);
    v += dpp_u<DPP_MIRROR>(v);
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), s1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t s2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), s3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return (s0 + s1) + (s2 + s3);
}

__device__ __forceinline__ float dist_f32(float ab, float b2, float rq, bool q_zero)
{
    // simsimd rules (oracle/semtools_oracle.c cos_finish): both zero -> 0,
    // ab == 0 -> 1, else max(0, 1 - ab * rsqrt(a2) * rsqrt(b2)).  rq = 0 when
    // the query is the zero vector, so "ab == 0 -> 1" falls out of the formula.
    if (b2 == 0.0f) return q_zero ? 0.0f : 1.0f;
    const float d = 1.0f - ab * rq * __frsqrt_rn(b2);
    return fmaxf(d, 0.0f);
}

// Read-only data produced by an EARLIER kernel, addressed uniformly: loads go through the scalar cache.
typedef const uint64_t __attribute__((address_space(4))) *const_u64_ptr;

// A value that is the same in every lane, moved to SGPRs (lets address arithmetic and loads go scalar).
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ key_t64 make_key(float d, uint32_t row)
{
    return ((key_t64)__float_as_uint(d) << 32) | (key_t64)row;
}


}  // namespace smt
