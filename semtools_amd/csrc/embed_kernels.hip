// embed_kernels.hip -- K1: model2vec static embedding on gfx950.
//
// Replaces the pool step of StaticModel::encode_with_args / encode_single
// (model2vec-rs 0.1.3; reference call sites src/search/mod.rs:69,138,153 and
// src/cmds/search.rs:136,154): for every line, gather the token rows of the
// f32 table [V x 256], sum them IN TOKEN ORDER in f32, divide by the token
// count, then divide by max(sqrt(sum_d v_d^2), 1e-12) with the squares summed
// IN DIMENSION ORDER -- i.e. exactly the serial f32 chains of the CPU code, so
// the output is bit-identical to the oracle (oracle/semtools_oracle.c
// orc_pool_ids), not merely close.
//
// Mapping: 16 lanes (one DPP row) own one line, 4 lines per wave.  Lane a of a
// row group owns dims {64c + 4a .. 64c + 4a + 3 : c = 0..3}: every gather
// instruction reads 16 lanes x 16 B = 256 contiguous bytes of a table row
// (4 instructions cover the 1 KiB row), and each lane's 16 accumulators are
// independent per-dimension chains, so token order is preserved trivially.
// The norm's dimension-order chain runs around the row group with DPP
// row_ror:1 (64 steps x 4 adds), all 4 lines of the wave in parallel.
#include "common.h"

namespace smt {

// row_ror:n rotates the 16-lane DPP row right by n: lane i receives lane (i - n) mod 16.
constexpr int DPP_ROW_ROR1 = 0x121;
template <int CTRL>
__device__ __forceinline__ float dppf(float v)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

struct EmbedParams {
    const float *table;
    uint64_t V;
    const uint32_t *ids;
    const uint64_t *offsets;
    uint64_t n_lines;
    uint32_t max_tokens;
    int normalize;
    float *out;
};

__global__ void __launch_bounds__(256) embed_kernel(EmbedParams p)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int a = lane & 15;        // position inside the row group
    const int g = lane >> 4;        // which of the wave's 4 lines
    const uint64_t wave_global = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t line = wave_global * 4 + g;
    const bool live = line < p.n_lines;

    uint64_t t0 = 0, n_tok = 0;
    if (live) {
        t0 = p.offsets[line];
        n_tok = p.offsets[line + 1] - t0;
        if (p.max_tokens != 0 && n_tok > (uint64_t)p.max_tokens) n_tok = p.max_tokens;
    }
    // the wave iterates to the longest of its 4 lines
    uint64_t n_max = n_tok;
    n_max = max(n_max, (uint64_t)__shfl_xor((unsigned long long)n_max, 16));
    n_max = max(n_max, (uint64_t)__shfl_xor((unsigned long long)n_max, 32));

    float4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);

    constexpr int TU = 4;  // tokens in flight per line
    for (uint64_t t = 0; t < n_max; t += TU) {
        float4 r[TU][4];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const bool on = (t + u) < n_tok;
            uint64_t id = on ? (uint64_t)p.ids[t0 + t + u] : 0;
            const bool ok = on && id < p.V;  // out-of-vocab ids contribute nothing
            const float4 *row = reinterpret_cast<const float4 *>(p.table + (ok ? id : 0) * 256);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                r[u][c] = ok ? row[c * 16 + a] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            if ((t + u) < n_tok) {  // token order: u ascending
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c].x = acc[c].x + r[u][c].x;
                    acc[c].y = acc[c].y + r[u][c].y;
                    acc[c].z = acc[c].z + r[u][c].z;
                    acc[c].w = acc[c].w + r[u][c].w;
                }
            }
        }
    }

    const float cnt = (float)(n_tok > 0 ? n_tok : 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        acc[c].x = acc[c].x / cnt; acc[c].y = acc[c].y / cnt;
        acc[c].z = acc[c].z / cnt; acc[c].w = acc[c].w / cnt;
    }

    if (p.normalize) {
        // ss = (((0 + v0^2) + v1^2) + ... + v255^2), dimension order.
        // step s = 16*c + a': the true chain value sits in lane a' of the group;
        // every lane runs the same instruction stream, only lane a' matters.
        float sq[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            sq[c][0] = acc[c].x * acc[c].x; sq[c][1] = acc[c].y * acc[c].y;
            sq[c][2] = acc[c].z * acc[c].z; sq[c][3] = acc[c].w * acc[c].w;
        }
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int step = 0; step < 16; ++step) {
                // take the running sum from the previous lane of the row (lane 15 -> lane 0 wraps
                // into the next 64-dim chunk); at the very first step everyone holds 0.
                const float in = (c == 0 && step == 0) ? 0.0f : dppf<DPP_ROW_ROR1>(s);
                s = (((in + sq[c][0]) + sq[c][1]) + sq[c][2]) + sq[c][3];
            }
        }
        // after 64 steps the full chain value is in lane 15 of each row group
        const float ss = __shfl(s, (lane & 48) | 15);
        float norm = sqrtf(ss);
        if (!(norm > 1e-12f)) norm = 1e-12f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[c].x = acc[c].x / norm; acc[c].y = acc[c].y / norm;
            acc[c].z = acc[c].z / norm; acc[c].w = acc[c].w / norm;
        }
    }

    if (live) {
        float4 *o = reinterpret_cast<float4 *>(p.out + line * 256);
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c * 16 + a] = acc[c];
    }
}

// ---- variant B: ONE WAVE PER LINE.  Lane l owns dims 4l .. 4l+3, so every gather instruction reads one whole table
// row -- 1 KiB contiguous, K2-style -- instead of four 256-B pieces of four different rows; 8 tokens in flight.  The
// per-dimension sums are the same serial f32 chains in token order; the norm's dimension-order chain walks the 64
// lanes with DPP wave_shr:1 (after step k lane k holds the chain through its own 4 dims), which costs the wave 64 x 5
// instructions for ONE line where variant A pays them for four -- the price of the coalesced gather.
constexpr int DPP_WAVE_SHR1_E = 0x138;
__global__ void __launch_bounds__(256) embed_wave_kernel(EmbedParams p)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const uint64_t line = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;   // wave-uniform
    if (line >= p.n_lines) return;
    const uint64_t t0 = p.offsets[line];
    uint64_t n_tok = p.offsets[line + 1] - t0;
    if (p.max_tokens != 0 && n_tok > (uint64_t)p.max_tokens) n_tok = p.max_tokens;

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int TU = 8;
    for (uint64_t t = 0; t < n_tok; t += TU) {
        float4 r[TU];
        bool ok[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const bool on = (t + u) < n_tok;                       // wave-uniform
            const uint64_t id = on ? (uint64_t)p.ids[t0 + t + u] : 0;
            ok[u] = on && id < p.V;                                // out-of-vocab ids contribute nothing
            r[u] = reinterpret_cast<const float4 *>(p.table + (ok[u] ? id : 0) * 256)[lane];
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            if ((t + u) < n_tok) {                                 // token order: u ascending; an invalid id adds +0
                const float4 v = ok[u] ? r[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
            }
        }
    }
    const float cnt = (float)(n_tok > 0 ? n_tok : 1);
    acc.x = acc.x / cnt; acc.y = acc.y / cnt; acc.z = acc.z / cnt; acc.w = acc.w / cnt;
    if (p.normalize) {
        const float s0 = acc.x * acc.x, s1 = acc.y * acc.y, s2 = acc.z * acc.z, s3 = acc.w * acc.w;
        float s = 0.0f;
#pragma unroll 16
        for (int step = 0; step < 64; ++step) {
            const float in = step == 0 ? 0.0f : dppf<DPP_WAVE_SHR1_E>(s);   // lane l takes lane l-1's running sum
            s = (((in + s0) + s1) + s2) + s3;
        }
        const float ss = __shfl(s, 63);
        float norm = sqrtf(ss);
        if (!(norm > 1e-12f)) norm = 1e-12f;
        acc.x = acc.x / norm; acc.y = acc.y / norm; acc.z = acc.z / norm; acc.w = acc.w / norm;
    }
    reinterpret_cast<float4 *>(p.out + line * 256)[lane] = acc;
}

int launch_embed(smt_ctx *ctx, const float *table, uint64_t V, int normalize, const uint32_t *ids,
                 const uint64_t *offsets, uint64_t n_lines, uint32_t max_tokens, float *out)
{
    if (n_lines == 0) return SMT_OK;
    EmbedParams p;
    p.table = table;
    p.V = V;
    p.ids = ids;
    p.offsets = offsets;
    p.n_lines = n_lines;
    p.max_tokens = max_tokens;
    p.normalize = normalize;
    p.out = out;
    const int threads = 256;
    if (ctx->tune.embed_wave_per_line) {           // 4 waves = 4 lines per block
        const uint64_t blocks = (n_lines + 3) / 4;
        SMT_REQUIRE(blocks < (1ull << 31), "too many lines for one embed launch");
        prof_begin(ctx, "embed");
        hipLaunchKernelGGL(embed_wave_kernel, dim3((unsigned)blocks), dim3(threads), 0, ctx->stream, p);
        prof_end(ctx, "embed");
        SMT_HIP_CHECK(hipGetLastError());
        return SMT_OK;
    }
    const uint64_t waves = (n_lines + 3) / 4;      // 4 waves = 16 lines per block
    const uint64_t blocks = (waves + 3) / 4;
    SMT_REQUIRE(blocks < (1ull << 31), "too many lines for one embed launch");
    prof_begin(ctx, "embed");
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)blocks), dim3(threads), 0, ctx->stream, p);
    prof_end(ctx, "embed");
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

}  // namespace smt
