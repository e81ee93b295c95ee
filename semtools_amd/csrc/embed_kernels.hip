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
// Mapping: 16 lanes (one DPP row) form a GROUP; lane a of a group owns dims {64c + 4a .. 64c + 4a + 3 : c = 0..3}:
// every gather instruction reads 16 lanes x 16 B = 256 contiguous bytes of a table row (4 instructions cover the
// 1 KiB row; tools/micro/gather_patterns.hip: this shape pulls random 1 KiB rows at 6.9 TB/s, the streaming ceiling of
// the part, also with the row number loaded from memory first), and each lane's 16 accumulators are independent
// per-dimension chains, so token order is preserved trivially.  The norm's dimension-order chain runs around the
// group with DPP row_ror:1 (64 steps x 4 adds).
//
// Work: a group owns a CONTIGUOUS RUN of lines and walks it on its own -- the four groups of a wave share an
// instruction stream but not a position: each step a group gathers the next 4 tokens of ITS current line, and a
// group whose line is complete finalises it (mean, norm, store) and moves on while the others keep gathering.  The
// first version gave a wave four lines and iterated to the longest of them: with ragged lines (0..32 tokens) a third of
// the gather slots sat idle and the kernel stopped at 5.3 TB/s on uniform ids (0.76 of what the probe shows the part
// delivers for this access pattern).
#include "common.h"

namespace smt {

// row_ror:n rotates the 16-lane DPP row right by n: lane i receives lane (i - n) mod 16.
constexpr int DPP_ROW_ROR1 = 0x121;
// 32-bit positions of the PF kernel step up to TU + 3 past a line's end before they are compared
constexpr uint64_t EMBED_SPAN_LIMIT = (1ull << 32) - 16;
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
    int batched;                // 1: parked lines wait for a wave-wide epilogue (0: each line is finished when it completes; A/B)
    float *out;
    uint64_t lines_per_group;   // a group of 16 lanes walks lines [g * lines_per_group, ...) -- unless:
    const uint64_t *run_start;  // [n_groups + 1] or nullptr: group g walks lines [run_start[g], run_start[g + 1]) (embed_runs_kernel)
    uint64_t n_groups;
    uint64_t span_limit;        // PF kernel: runs spanning this many tokens or more are left to the generic kernel (2^32; tests: small)
    int only_large;             // generic kernel: 1 = walk only the runs the PF kernel left
};

// PF: the token ids of step s + 1 are requested while the rows of step s are in flight (the next position is pure arithmetic on the
// prefetched offsets), so a step waits for ONE memory round trip -- the rows -- instead of two in a row (ids, then rows).
// The PF kernel keeps token positions as 32-bit counts from the first token of its run (six position registers instead of twelve:
// that is what takes it under the 128-VGPR line of four waves per SIMD without spilling); a run spanning 2^32 tokens or more --
// 16 GiB of ids under one group -- is skipped here and walked by the generic kernel, launched behind it with only_large = 1.
// Runs of equal WORK instead of equal line counts.  A group's time is its tokens plus ~4 token-steps per line (the epilogue), and
// line lengths of real text have a heavy tail: with equal line counts a run that holds one 2048-token line among four-token ones
// takes three times as long as its neighbours and the launch waits for it (8 M lines of 4 tokens with 0.2 % of 2048: 6.6 ms against
// 4.7 for the same tokens spread evenly).  key(i) = offsets[i] - offsets[0] + 4 i is strictly increasing; run_start[g] = the first
// line whose key reaches g / n_groups of key(n_lines) -- a binary search per group, one thread each.  (The truncation to
// max_tokens is not in the key: a line far beyond it spreads its share over groups that then find nothing to do.)
constexpr uint64_t EMBED_LINE_WEIGHT = 4;
__global__ void __launch_bounds__(256) embed_runs_kernel(const uint64_t *__restrict__ offsets, uint64_t n_lines, uint64_t n_groups,
                                                         uint64_t *__restrict__ run_start)
{
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g > n_groups) return;
    const uint64_t o0 = offsets[0];
    const uint64_t total = offsets[n_lines] - o0 + EMBED_LINE_WEIGHT * n_lines;
    // target = total * g / n_groups without overflow: total < 2^52 in practice, g <= 2^20 -- 128-bit to be safe
    const uint64_t target = (uint64_t)(((unsigned __int128)total * g) / n_groups);
    uint64_t lo = 0, hi = n_lines;   // first i in [0, n_lines] with key(i) >= target (key(n_lines) = total >= target)
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (offsets[mid] - o0 + EMBED_LINE_WEIGHT * mid >= target) hi = mid; else lo = mid + 1;
    }
    run_start[g] = g == n_groups ? n_lines : lo;
}

template <bool PF>
__global__ void __launch_bounds__(256, 4) embed_kernel(EmbedParams p)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int a = lane & 15;        // position inside the group
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    // (register diet: the kernel sits at the 128-VGPR edge of four waves per SIMD -- the position inside the run is a 32-bit count,
    // the parked line is always the one before the current, its token count is kept as the float the mean divides by)
    uint64_t line0;
    uint32_t n_run;
    if (p.run_start) {   // (kernel-uniform)
        const bool mine = group < p.n_groups;
        line0 = mine ? p.run_start[group] : 0;
        n_run = mine ? (uint32_t)(p.run_start[group + 1] - line0) : 0u;
    } else {
        line0 = group * p.lines_per_group;
        n_run = line0 < p.n_lines ? (uint32_t)min(p.n_lines - line0, p.lines_per_group) : 0u;
    }
    // positions are counted from `tb`: the run's first token in the PF kernel, 0 in the generic one
    using pos_t = typename std::conditional<PF, uint32_t, uint64_t>::type;
    uint64_t tb = 0;
    if (n_run != 0) {
        const uint64_t first = p.offsets[line0];
        const bool large = p.offsets[line0 + n_run] - first >= p.span_limit;
        if constexpr (PF) {
            tb = first;
            if (large) n_run = 0;
        } else if (p.only_large && !large) n_run = 0;
    }
    uint32_t li = 0;                                     // current line = line0 + li
    bool active = li < n_run;                            // group-uniform

    // token range of the current line [t, t_end) (max_tokens applied), its token count, and the NEXT line's end (read one
    // line ahead: a group that moves on must not wait for a dependent offsets load)
    pos_t t = 0, t_end = 0, o_end = 0, o_next = 0;
    if (active) {
        t = (pos_t)(p.offsets[line0] - tb);
        o_end = (pos_t)(p.offsets[line0 + 1] - tb);
        o_next = 1 < n_run ? (pos_t)(p.offsets[line0 + 2] - tb) : o_end;
        t_end = o_end;
        if (p.max_tokens != 0 && t_end - t > (pos_t)p.max_tokens) t_end = t + p.max_tokens;
    }
    float cnt_cur = (float)(t_end - t > 0 ? t_end - t : 1);   // what the current line's sums are divided by (max(cnt, 1))

    float4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);

    float4 pend[4];                  // the sums of a complete line waiting for its epilogue
    float pend_cnt = 1.0f;
    bool has_pend = false;           // group-uniform; the parked line is line0 + li - 1
#pragma unroll
    for (int c = 0; c < 4; ++c) pend[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto finalize = [&]() __attribute__((always_inline)) {
        if (has_pend) {
            const float cnt = pend_cnt;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                pend[c].x = pend[c].x / cnt; pend[c].y = pend[c].y / cnt;
                pend[c].z = pend[c].z / cnt; pend[c].w = pend[c].w / cnt;
            }
            if (p.normalize) {
                // ss = (((0 + v0^2) + v1^2) + ... + v255^2), dimension order.
                // step s = 16*c + a': the true chain value sits in lane a' of the group;
                // every lane runs the same instruction stream, only lane a' matters.
                float sq[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    sq[c][0] = pend[c].x * pend[c].x; sq[c][1] = pend[c].y * pend[c].y;
                    sq[c][2] = pend[c].z * pend[c].z; sq[c][3] = pend[c].w * pend[c].w;
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
                // after 64 steps the full chain value is in lane 15 of each group
                const float ss = __shfl(s, (lane & 48) | 15);
                float norm = sqrtf(ss);
                if (!(norm > 1e-12f)) norm = 1e-12f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    pend[c].x = pend[c].x / norm; pend[c].y = pend[c].y / norm;
                    pend[c].z = pend[c].z / norm; pend[c].w = pend[c].w / norm;
                }
            }
            float4 *o = reinterpret_cast<float4 *>(p.out + (line0 + li - 1) * 256);
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c * 16 + a] = pend[c];
            has_pend = false;
        }
    };

    constexpr int TU = 4;  // tokens in flight per group
    // PF: the ids of the step about to run -- ONE per lane (lane a holds token a & 3 of its group's step; the gather reads it with a
    // DPP quad broadcast), so the prefetch costs one register and a quarter of the id loads
    static_assert(TU == 4, "one id per lane of a quad");
    uint32_t nid = 0;
    const uint32_t *ids = p.ids + tb;
    if constexpr (PF) nid = (active && (t + (a & 3)) < t_end) ? ids[t + (a & 3)] : 0u;
    while (__any(active)) {
        float4 r[TU][4];
        uint32_t idq[TU] = {0u, 0u, 0u, 0u};
        if constexpr (PF) {   // quad_perm broadcasts of lane u of every quad
            idq[0] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nid, 0x00, 0xF, 0xF, false);
            idq[1] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nid, 0x55, 0xF, 0xF, false);
            idq[2] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nid, 0xAA, 0xF, 0xF, false);
            idq[3] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nid, 0xFF, 0xF, 0xF, false);
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const bool on = active && (t + u) < t_end;
            uint64_t id;
            if constexpr (PF) id = idq[u];
            else id = on ? (uint64_t)ids[t + u] : 0;
            const bool ok = on && id < p.V;  // out-of-vocab ids contribute nothing
            const float4 *row = reinterpret_cast<const float4 *>(p.table + (ok ? id : 0) * 256);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // (plain loads, not nontemporal: natural text is Zipf-distributed and its hot rows must stay in L2 / MALL --
                // measured with the nt policy: Zipf ids 3.50 -> 4.59 ms, uniform ids 5.97 -> 6.27 ms; round 4: 3.18 -> 4.68, 6.56 -> 6.77)
                r[u][c] = ok ? row[c * 16 + a] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // where the group stands after this step -- known before any row has arrived
        const bool done = active && t + TU >= t_end;
        if constexpr (PF) {
            const bool n_active = done ? li + 1 < n_run : active;
            const pos_t nt = done ? o_end : t + TU;
            pos_t nt_end = done ? o_next : t_end;
            if (done && p.max_tokens != 0 && nt_end - nt > (pos_t)p.max_tokens) nt_end = nt + p.max_tokens;
            nid = (n_active && (nt + (a & 3)) < nt_end) ? ids[nt + (a & 3)] : 0u;
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            if (active && (t + u) < t_end) {  // token order: u ascending
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c].x = acc[c].x + r[u][c].x;
                    acc[c].y = acc[c].y + r[u][c].y;
                    acc[c].z = acc[c].z + r[u][c].z;
                    acc[c].w = acc[c].w + r[u][c].w;
                }
            }
        }
        t += TU;
        // ---- a complete line is PARKED (its sums move to `pend`) and the group goes on gathering its next line; the epilogue --
        // mean, norm chain, store: ~640 instructions, the bulk of a line's instruction count -- runs for all parked lines of the
        // wave at once: when every group has one parked, or when a group completes a second line before that.  (Run at once per
        // completion it executed with 16 of 64 lanes enabled, once per line; measured: Zipf ids 3.75 -> 3.39 ms)
        if (__any(done && has_pend)) finalize();
        if (done) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                pend[c] = acc[c];
                acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            pend_cnt = cnt_cur;
            has_pend = true;
            // ---- next line of the run
            ++li;
            active = li < n_run;
            t = o_end;
            o_end = o_next;
            if (li + 1 < n_run) o_next = (pos_t)(p.offsets[line0 + li + 2] - tb);
            t_end = o_end;
            if (p.max_tokens != 0 && t_end - t > (pos_t)p.max_tokens) t_end = t + p.max_tokens;
            cnt_cur = (float)(t_end - t > 0 ? t_end - t : 1);
        }
        if (__any(has_pend) && (!p.batched || __all(has_pend || !active))) finalize();
    }
    if (__any(has_pend)) finalize();
}

int launch_embed(smt_ctx *ctx, const float *table, uint64_t V, int normalize, const uint32_t *ids,
                 const uint64_t *offsets, uint64_t n_lines, uint32_t max_tokens, float *out, uint64_t n_tokens_known)
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
    p.batched = ctx->tune.embed_batched & 1;
    p.out = out;
    // a run of lines per group: enough groups to fill the chip (16 waves x 4 groups per CU), runs long enough that ragged
    // lines average out inside a run (a group with 100 lines of 0..32 tokens ends within ~5 % of its neighbours)
    const uint64_t want_groups = (uint64_t)std::max(ctx->num_cus, 1) * 64;
    p.lines_per_group = std::max<uint64_t>(1, (n_lines + want_groups - 1) / want_groups);
    uint64_t groups = (n_lines + p.lines_per_group - 1) / p.lines_per_group;
    p.run_start = nullptr;
    p.n_groups = 0;
    // more than one line per group: the runs are cut by work, not by line count (embed_runs_kernel; tuning key embed_batched bit 3
    // turns it off for A/B)
    const bool balanced = p.lines_per_group > 1 && !(ctx->tune.embed_batched & 8);
    if (balanced) {
        groups = want_groups;
        if (ctx->embed_runs_cap < groups + 1) {
            SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (ctx->d_embed_runs) SMT_HIP_CHECK(hipFree(ctx->d_embed_runs));
            ctx->d_embed_runs = nullptr;
            ctx->embed_runs_cap = 0;
            SMT_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_embed_runs), (groups + 1) * sizeof(uint64_t)));
            ctx->embed_runs_cap = groups + 1;
        }
        p.run_start = ctx->d_embed_runs;
        p.n_groups = groups;
    }
    const uint64_t blocks = (groups + 15) / 16;    // 256 threads = 16 groups
    SMT_REQUIRE(blocks < (1ull << 24), "too many lines for one embed launch");
    prof_begin(ctx, "embed");
    if (balanced)
        hipLaunchKernelGGL(embed_runs_kernel, dim3((unsigned)((groups + 1 + 255) / 256)), dim3(256), 0, ctx->stream, offsets, n_lines, groups,
                           ctx->d_embed_runs);
    // (A/B: embed_batched bit 1 = ids prefetched one step ahead.  Probed in round 4 and removed: the same kernel at three waves per
    // SIMD without spills -- Zipf 3.11 -> 3.32 ms -- and nontemporal row loads -- 4.68 / 6.77 ms; profiles/r04_k1/)
    // embed_batched bit 2 (tests): the PF kernel leaves every run of 64 tokens or more to the generic kernel
    p.span_limit = (ctx->tune.embed_batched & 4) ? 64ull : EMBED_SPAN_LIMIT;
    p.only_large = 0;
    if (ctx->tune.embed_batched & 2) {
        hipLaunchKernelGGL(embed_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p);
        // the runs it left: none when the caller knows the batch holds fewer tokens than the limit (n_tokens_known = 0: offsets
        // only exist on the device -- the launch finds nothing to do and costs a few microseconds)
        if (n_tokens_known == 0 || n_tokens_known >= p.span_limit) {
            p.only_large = 1;
            hipLaunchKernelGGL(embed_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p);
        }
    } else hipLaunchKernelGGL(embed_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p);
    prof_end(ctx, "embed");
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

}  // namespace smt
