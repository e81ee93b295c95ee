// api.cpp -- host side of libsemtools_hip.so: error text, contexts, tuning keys, models, corpora (with their fp16 operand image),
// embed, ids -- the extern "C" entry points of include/semtools_hip.h that are not searches (search.cpp), files (corpus_io.cpp),
// groups (group.cpp, sharded.cpp) or indexes (ivfpq_*.hip).  There is no CPU fallback anywhere: every compute entry point needs
// a live gfx950 context.
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <limits>
#include <string>

#include <unistd.h>

#include "common.h"

namespace smt {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ensure_scratch(smt_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_bytes) return SMT_OK;
    // stream-ordered safety: earlier kernels (main stream, async selects on the aux stream) may still read the old buffer
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->aux_stream) SMT_HIP_CHECK(hipStreamSynchronize(ctx->aux_stream));
    if (ctx->d_scratch) SMT_HIP_CHECK(hipFree(ctx->d_scratch));
    ctx->d_scratch = nullptr;
    ctx->scratch_bytes = 0;
    size_t want = std::max(bytes, (size_t)1 << 20);
    SMT_HIP_CHECK(hipMalloc(&ctx->d_scratch, want));
    ctx->scratch_bytes = want;
    return SMT_OK;
}

int ensure_async(smt_ctx *ctx)
{
    if (ctx->aux_stream) return SMT_OK;
    SMT_HIP_CHECK(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    SMT_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->d_flags), 64));
    SMT_HIP_CHECK(hipMemsetAsync(ctx->d_flags, 0, 64, ctx->stream));
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->async_step = 0;
    return SMT_OK;
}

int drain_async(smt_ctx *ctx)
{
    if (!ctx->async_pending) return SMT_OK;
    ctx->async_pending = false;
    if (!ctx->aux_stream || !ctx->d_flags) return SMT_OK;
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));      // the scans the selects are waiting for
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->aux_stream));
    unsigned long long timed_out = 0;
    SMT_HIP_CHECK(hipMemcpy(&timed_out, ctx->d_flags + 3, sizeof(timed_out), hipMemcpyDeviceToHost));
    if (timed_out) {
        (void)hipMemset(ctx->d_flags + 3, 0, sizeof(timed_out));
        set_error("async select: a kernel gave up waiting for its partner (flag wait timed out)");
        return SMT_E_HIP;
    }
    return SMT_OK;
}

int ensure_stage(smt_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->stage_bytes) return SMT_OK;
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->d_stage) SMT_HIP_CHECK(hipFree(ctx->d_stage));
    ctx->d_stage = nullptr;
    ctx->stage_bytes = 0;
    size_t want = std::max(bytes, (size_t)1 << 16);
    SMT_HIP_CHECK(hipMalloc(&ctx->d_stage, want));
    ctx->stage_bytes = want;
    return SMT_OK;
}

int ensure_pinned(smt_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->pinned_bytes) return SMT_OK;
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->h_pinned) SMT_HIP_CHECK(hipHostFree(ctx->h_pinned));
    ctx->h_pinned = nullptr;
    ctx->pinned_bytes = 0;
    size_t want = std::max(bytes, (size_t)1 << 16);
    SMT_HIP_CHECK(hipHostMalloc(&ctx->h_pinned, want, hipHostMallocDefault));
    ctx->pinned_bytes = want;
    return SMT_OK;
}

int ensure_pinned_in(smt_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->pinned_in_bytes) return SMT_OK;
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->h_pinned_in) SMT_HIP_CHECK(hipHostFree(ctx->h_pinned_in));
    ctx->h_pinned_in = nullptr;
    ctx->pinned_in_bytes = 0;
    size_t want = std::max(bytes, (size_t)1 << 16);
    SMT_HIP_CHECK(hipHostMalloc(&ctx->h_pinned_in, want, hipHostMallocDefault));
    ctx->pinned_in_bytes = want;
    return SMT_OK;
}

void prof_begin_on(smt_ctx *ctx, const char *name, hipStream_t st)
{
    if (!ctx->prof_on) return;
    ProfEntry &e = ctx->prof[name];
    const uint64_t every = ctx->tune.prof_every > 1 ? (uint64_t)ctx->tune.prof_every : 1;
    e.armed = (e.calls++ % every) == 0;
    if (!e.armed) return;
    if (e.used + 2 > e.ev.size()) {
        const size_t old = e.ev.size();
        e.ev.resize(old + 256);
        for (size_t i = old; i < e.ev.size(); ++i) (void)hipEventCreate(&e.ev[i]);
    }
    (void)hipEventRecord(e.ev[e.used], st);
}

void prof_end_on(smt_ctx *ctx, const char *name, hipStream_t st)
{
    if (!ctx->prof_on) return;
    ProfEntry &e = ctx->prof[name];
    if (!e.armed) return;
    e.armed = false;
    (void)hipEventRecord(e.ev[e.used + 1], st);
    e.used += 2;
}

void prof_begin(smt_ctx *ctx, const char *name) { prof_begin_on(ctx, name, ctx->stream); }
void prof_end(smt_ctx *ctx, const char *name) { prof_end_on(ctx, name, ctx->stream); }

int check_ctx(const smt_ctx *ctx)
{
    if (!ctx) { set_error("null context"); return SMT_E_INVALID; }
    return SMT_OK;
}

int api_catch() noexcept
{
    try { throw; }
    catch (const std::bad_alloc &) {
        try { set_error("out of host memory"); } catch (...) {}
        return SMT_E_NOMEM;
    } catch (const std::exception &e) {
        try { set_error("%s", e.what()); } catch (...) {}
        return SMT_E_INVALID;
    } catch (...) {
        try { set_error("unknown C++ exception"); } catch (...) {}
        return SMT_E_INVALID;
    }
}

int bind_device(smt_ctx *ctx, bool drain)
{
    SMT_HIP_CHECK(hipSetDevice(ctx->device));
    return drain ? drain_async(ctx) : SMT_OK;
}

// Device buffers carved out of one temporary allocation, freed on scope exit.
struct DeviceTemp {
    void *p = nullptr;
    ~DeviceTemp() { if (p) (void)hipFree(p); }
};

static uint64_t fnv1a(const uint8_t *b, uint64_t n)
{
    // reference src/workspace/store.rs:651-661
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint64_t i = 0; i < n; ++i) { h ^= (uint64_t)b[i]; h *= 0x100000001b3ull; }
    return h;
}

int corpus_reserve(smt_corpus *c, uint64_t rows_needed)
{
    if (rows_needed <= c->capacity) return SMT_OK;
    corpus_writer_drain(c);   // the rows are about to move: a write-ahead job reads them where they are
    if (!c->owned) { set_error("corpus adopted from device memory cannot grow"); return SMT_E_NOMEM; }
    uint64_t cap = std::max<uint64_t>(c->capacity * 2, rows_needed);
    cap = std::max<uint64_t>(cap, 1024);
    float *nd = nullptr;
    const size_t bytes = (size_t)cap * c->dim * sizeof(float);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&nd), bytes);
    if (e != hipSuccess && c->image) {
        // the operand image is derived data (up to as many bytes as the rows): the rows' growth goes first.  A later batch builds
        // it again if there is room then (image_mode stays "by policy").
        (void)hipGetLastError();
        (void)hipStreamSynchronize(c->ctx->stream);
        corpus_image_drop(c);
        e = hipMalloc(reinterpret_cast<void **>(&nd), bytes);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc(%zu) for corpus rows: %s", bytes, hipGetErrorString(e)); return SMT_E_NOMEM; }
    if (c->rows) {
        e = hipMemcpyAsync(nd, c->d_rows, (size_t)c->rows * c->dim * sizeof(float), hipMemcpyDeviceToDevice, c->ctx->stream);
        if (e != hipSuccess) { (void)hipFree(nd); set_error("corpus grow copy: %s", hipGetErrorString(e)); return SMT_E_HIP; }
    }
    SMT_HIP_CHECK(hipStreamSynchronize(c->ctx->stream));
    if (c->d_rows) (void)hipFree(c->d_rows);
    c->d_rows = nd;
    c->capacity = cap;
    return SMT_OK;
}

void corpus_image_drop(smt_corpus *c)
{
    if (c->image) (void)hipFree(c->image);
    if (c->image_zero) (void)hipFree(c->image_zero);
    c->image = nullptr;
    c->image_zero = nullptr;
    c->image_cap_tiles = 0;
    c->image_rows = 0;
}

// Policy: an image is kept for a corpus whose memory this library owns (every change of a row goes through this file) from the
// first batch of >= 8 queries over >= 64 Ki rows on (tuning key corpus_image), and for any corpus it was asked for
// (smt_corpus_prepack: the caller of smt_corpus_from_device then answers for the rows not changing behind the image).
// It costs half the corpus again in HBM; when that allocation fails the corpus goes on without one.
int corpus_image_sync(smt_corpus *c, uint32_t nq, const void **image, const uint32_t **image_zero)
{
    *image = nullptr;
    *image_zero = nullptr;
    if (c->image_mode < 0) return SMT_OK;
    if (c->image_mode == 0 && !(c->owned && c->ctx->tune.corpus_image != 0 && (c->image || (nq >= 8 && c->rows >= 65536)))) return SMT_OK;
    if (c->rows == 0) return SMT_OK;
    const uint64_t tiles = (c->rows + 31) / 32;
    if (tiles > c->image_cap_tiles) {
        // room to grow: x 1.5 of what it held, or what the rows have reserved (at most twice the rows that exist)
        const uint64_t cap = std::max<uint64_t>(tiles, c->image ? c->image_cap_tiles + c->image_cap_tiles / 2
                                                                 : std::min<uint64_t>((c->capacity + 31) / 32, 2 * tiles));
        void *ni = nullptr;
        uint32_t *nz = nullptr;
        if (hipMalloc(&ni, (size_t)cap * 16384) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&nz), (size_t)cap * 4) != hipSuccess) {
            (void)hipGetLastError();
            if (ni) (void)hipFree(ni);
            corpus_image_drop(c);
            c->image_mode = -1;   // no room beside the rows: this corpus is searched from its f32 rows
            return SMT_OK;
        }
        const uint64_t keep = c->image ? c->image_rows / 32 : 0;   // whole tiles that stay valid
        if (keep) {
            hipError_t e = hipMemcpyAsync(ni, c->image, (size_t)keep * 16384, hipMemcpyDeviceToDevice, c->ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(nz, c->image_zero, (size_t)keep * 4, hipMemcpyDeviceToDevice, c->ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->ctx->stream);
            if (e != hipSuccess) {
                (void)hipFree(ni);
                (void)hipFree(nz);
                set_error("operand image grow copy: %s", hipGetErrorString(e));
                return SMT_E_HIP;
            }
        }
        const uint64_t had = c->image_rows;
        corpus_image_drop(c);
        c->image = ni;
        c->image_zero = nz;
        c->image_cap_tiles = cap;
        c->image_rows = keep * 32 <= had ? keep * 32 : 0;
    }
    if (c->image_rows < c->rows) {
        const uint64_t first = c->image_rows / 32;   // a partly filled last tile is packed again
        const int rc = launch_pack_image(c->ctx, c->d_rows, c->rows, first, tiles - first, c->image, c->image_zero);
        if (rc) return rc;
        c->image_rows = c->rows;
    }
    *image = c->image;
    *image_zero = c->image_zero;
    return SMT_OK;
}

}  // namespace smt

using namespace smt;

extern "C" {

const char *smt_last_error(void) { return g_err; }
const char *smt_version(void) { return "semtools-hip 0.1.0 (gfx950)"; }

int smt_device_count(void)
try {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return SMT_E_HIP; }
    return n;
} catch (...) { return smt::api_catch(); }

static int ctx_create_impl(int device, void *stream, bool use_given, smt_ctx **out)
{
    SMT_REQUIRE(out != nullptr, "out");
    *out = nullptr;
    int n = 0;
    SMT_HIP_CHECK(hipGetDeviceCount(&n));
    if (n <= 0) { set_error("no HIP device visible: libsemtools_hip has no CPU fallback"); return SMT_E_HIP; }
    SMT_REQUIRE(device >= 0 && device < n, "device index out of range");
    SMT_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    SMT_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return SMT_E_HIP;
    }
    smt_ctx *ctx = new (std::nothrow) smt_ctx();
    if (!ctx) { set_error("out of host memory"); return SMT_E_NOMEM; }
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    if (use_given) {
        ctx->stream = reinterpret_cast<hipStream_t>(stream);
        ctx->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; set_error("hipStreamCreate: %s", hipGetErrorString(e)); return SMT_E_HIP; }
        ctx->own_stream = true;
    }
    // sticky counter of selects whose exactness certificate failed (SelectArgs::f32_err)
    hipError_t es = hipMalloc(reinterpret_cast<void **>(&ctx->d_status), 64);
    if (es == hipSuccess) es = hipMemset(ctx->d_status, 0, 64);
    if (es != hipSuccess) {
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        set_error("context status word: %s", hipGetErrorString(es));
        return SMT_E_HIP;
    }
    *out = ctx;
    return SMT_OK;
}

int smt_ctx_create(int device, smt_ctx **out)
try {
    return ctx_create_impl(device, nullptr, false, out);
} catch (...) { return smt::api_catch(); }

int smt_ctx_create_on_stream(int device, void *stream, smt_ctx **out)
try {
    return ctx_create_impl(device, stream, true, out);
} catch (...) { return smt::api_catch(); }

void smt_ctx_destroy(smt_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
    if (ctx->d_flags) (void)hipFree(ctx->d_flags);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->d_steal) (void)hipFree(ctx->d_steal);
    for (auto &kv : ctx->prof)
        for (hipEvent_t ev : kv.second.ev) (void)hipEventDestroy(ev);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_embed_runs) (void)hipFree(ctx->d_embed_runs);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->h_pinned_in) (void)hipHostFree(ctx->h_pinned_in);
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int smt_ctx_synchronize(smt_ctx *ctx)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    SMT_HIP_CHECK(hipSetDevice(ctx->device));
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return drain_async(ctx);
} catch (...) { return smt::api_catch(); }

int smt_ctx_uncertain_count(smt_ctx *ctx, uint64_t *count, int reset)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    SMT_REQUIRE(count != nullptr, "null argument");
    if ((rc = bind_device(ctx))) return rc;
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    unsigned long long v = 0;
    SMT_HIP_CHECK(hipMemcpy(&v, ctx->d_status, sizeof(v), hipMemcpyDeviceToHost));
    if (reset && v) SMT_HIP_CHECK(hipMemset(ctx->d_status, 0, sizeof(v)));
    *count = v;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_ctx_aux_stream(smt_ctx *ctx, void **stream_out)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    SMT_REQUIRE(stream_out != nullptr, "null argument");
    SMT_HIP_CHECK(hipSetDevice(ctx->device));
    if ((rc = ensure_async(ctx))) return rc;
    *stream_out = reinterpret_cast<void *>(ctx->aux_stream);
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_prof_enable(smt_ctx *ctx, int on)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    ctx->prof_on = on != 0;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_prof_reset(smt_ctx *ctx)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (auto &kv : ctx->prof) { kv.second.used = 0; kv.second.calls = 0; kv.second.armed = false; }
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_prof_read(smt_ctx *ctx, const char *kernel, uint64_t *launches, double *total_ms)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    SMT_REQUIRE(kernel && launches && total_ms, "null argument");
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->aux_stream) SMT_HIP_CHECK(hipStreamSynchronize(ctx->aux_stream));   // ("exchange" / "merge" pairs may lie on it)
    *launches = 0;
    *total_ms = 0.0;
    auto it = ctx->prof.find(kernel);
    if (it == ctx->prof.end()) return SMT_OK;
    ProfEntry &e = it->second;
    for (size_t i = 0; i + 1 < e.used; i += 2) {
        float ms = 0.f;
        SMT_HIP_CHECK(hipEventElapsedTime(&ms, e.ev[i], e.ev[i + 1]));
        *total_ms += (double)ms;
        *launches += 1;
    }
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_set_tuning(smt_ctx *ctx, const char *key, int64_t value)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    SMT_REQUIRE(key != nullptr, "key");
    std::string k(key);
    if (k == "scan_blocks") ctx->tune.scan_blocks = (int)value;
    else if (k == "scan_threads") {
        SMT_REQUIRE(value >= 64 && value <= 1024 && value % 64 == 0, "scan_threads must be a multiple of 64 in [64,1024]");
        ctx->tune.scan_threads = (int)value;
    } else if (k == "scan_unroll") {
        SMT_REQUIRE(value == 2 || value == 4 || value == 8, "scan_unroll must be 2, 4 or 8");   // (16 spilled 40 VGPRs and never won a sweep: removed)
        ctx->tune.scan_unroll = (int)value;
    } else if (k == "scan_nontemporal") ctx->tune.scan_nontemporal = (int)value;
    else if (k == "scan_prefetch") ctx->tune.scan_prefetch = (int)value;
    else if (k == "scan_steal") {
        SMT_REQUIRE(value == 0 || value == 1 || value == 2 || value == 4 || value == 8 || value == 16, "scan_steal: rounds per dynamically dealt group, a power of two up to 16 (0 = the static deal)");
        ctx->tune.scan_steal = (int)value;
    } else if (k == "scan_steal_pct") {
        SMT_REQUIRE(value >= 1 && value <= 50, "scan_steal_pct: 1..50 per cent of the corpus dealt dynamically");
        ctx->tune.scan_steal_pct = (int)value;
    }
    else if (k == "gemm_blocks") ctx->tune.gemm_blocks = (int)value;
    else if (k == "gemm_ldsrow") ctx->tune.gemm_ldsrow = (int)value;
    else if (k == "gemm_bootstrap") ctx->tune.gemm_bootstrap = (int)value;
    else if (k == "gemm_bf16x3") ctx->tune.gemm_bf16x3 = (int)value;
    else if (k == "gemm_rowreg") ctx->tune.gemm_rowreg = (int)value;
    else if (k == "gemm_nominate") ctx->tune.gemm_nominate = (value >= 1 && value <= 3) ? (int)value : 0;
    else if (k == "fallback_batch_min_rows") ctx->tune.fallback_batch_min_rows = value < 0 ? 0 : value;
    else if (k == "guard_band") ctx->tune.guard_band = (int)std::max<int64_t>(8, std::min<int64_t>(56, value));
    else if (k == "gemm_min_nq") ctx->tune.gemm_min_nq = (int)std::max<int64_t>(2, std::min<int64_t>(8, value));
    else if (k == "gemm_min_rows_small") ctx->tune.gemm_min_rows_small = value < 0 ? 0 : value;
    else if (k == "gemm_dma_nt") ctx->tune.gemm_dma_nt = (int)value;
    else if (k == "gemm_qsplit") ctx->tune.gemm_qsplit = (int)value;
    else if (k == "gemm_buffered") ctx->tune.gemm_buffered = (int)value;
    else if (k == "gemm_split_last") {
        SMT_REQUIRE(value >= 0 && value <= 2, "gemm_split_last: 0 no level in two parts, 1 a ratio-16 last level, 2 (default) also the first level after the bootstrap");
        ctx->tune.gemm_split_last = (int)value;
    } else if (k == "embed_batched") {
        SMT_REQUIRE(value >= 0 && value <= 15, "embed_batched is a bit mask: 1 batched id loads, 2 id prefetch kernel, 4 (tests) 64-token spans, 8 runs of equal line counts (A/B)");
        ctx->tune.embed_batched = (int)value;
    } else if (k == "gemm_resident") { /* the kernel it selected left in round 2: accepted and ignored, as before round 4 */ }
    else if (k == "gemm_image") ctx->tune.gemm_image = (int)value;
    else if (k == "corpus_image") ctx->tune.corpus_image = (int)value;
    else if (k == "image_scan_min_rows") ctx->tune.image_scan_min_rows = value < 0 ? 0 : value;
    else if (k == "gemm_boot_fine") ctx->tune.gemm_boot_fine = (int)value;
    else if (k == "image_use_min_rows") ctx->tune.image_use_min_rows = value < 0 ? 0 : value;
    else if (k == "prof_every") ctx->tune.prof_every = (int)value;
    else if (k == "merge_on_aux") {
        int rc2 = drain_async(ctx);
        if (rc2) return rc2;
        ctx->tune.merge_on_aux = (int)value;
    }
    else if (k == "async_select") {
        int rc2 = drain_async(ctx);
        if (rc2) return rc2;
        ctx->tune.async_select = (int)value;
    }
    else if (k == "scan_debug_ptr") ctx->tune.scan_debug_ptr = value;
    else if (k == "select_debug_ptr") ctx->tune.select_debug_ptr = value;
    else if (k == "prof_select") ctx->tune.prof_select = (int)value;
    else if (k == "direct_delivery") ctx->tune.direct_delivery = (int)value;
    else { set_error("unknown tuning key '%s'", key); return SMT_E_INVALID; }
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

/* ---------------------------------------------------------------- model ---- */

int smt_model_create(smt_ctx *ctx, const float *table_host, uint64_t V, uint32_t D, int normalize,
                     smt_model **out)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    SMT_REQUIRE(out && table_host, "null argument");
    *out = nullptr;
    if (D != SMT_DIM) { set_error("embedding dim %u unsupported (kernels are specialised for 256)", D); return SMT_E_UNSUPPORTED; }
    SMT_REQUIRE(V > 0, "empty table");
    if ((rc = bind_device(ctx))) return rc;
    smt_model *m = new (std::nothrow) smt_model();
    if (!m) { set_error("out of host memory"); return SMT_E_NOMEM; }
    m->ctx = ctx; m->V = V; m->D = D; m->normalize = normalize ? 1 : 0; m->owned = true;
    const size_t bytes = (size_t)V * D * sizeof(float);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&m->d_table), bytes);
    if (e != hipSuccess) { delete m; set_error("hipMalloc(%zu) for the embedding table: %s", bytes, hipGetErrorString(e)); return SMT_E_NOMEM; }
    e = hipMemcpyAsync(m->d_table, table_host, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(m->d_table); delete m; set_error("table upload: %s", hipGetErrorString(e)); return SMT_E_HIP; }
    *out = m;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_model_create_from_device(smt_ctx *ctx, const float *table_dev, uint64_t V, uint32_t D,
                                 int normalize, smt_model **out)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    SMT_REQUIRE(out && table_dev, "null argument");
    *out = nullptr;
    if (D != SMT_DIM) { set_error("embedding dim %u unsupported", D); return SMT_E_UNSUPPORTED; }
    smt_model *m = new (std::nothrow) smt_model();
    if (!m) { set_error("out of host memory"); return SMT_E_NOMEM; }
    m->ctx = ctx; m->V = V; m->D = D; m->normalize = normalize ? 1 : 0;
    m->d_table = const_cast<float *>(table_dev);
    m->owned = false;
    *out = m;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

void smt_model_destroy(smt_model *model)
{
    if (!model) return;
    (void)hipSetDevice(model->ctx->device);
    (void)hipStreamSynchronize(model->ctx->stream);
    if (model->owned && model->d_table) (void)hipFree(model->d_table);
    delete model;
}

/* --------------------------------------------------------------- corpus ---- */

int smt_corpus_create(smt_ctx *ctx, uint32_t D, uint64_t capacity_rows, smt_corpus **out)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    SMT_REQUIRE(out != nullptr, "out");
    *out = nullptr;
    if (D != SMT_DIM) { set_error("embedding dim %u unsupported (kernels are specialised for 256)", D); return SMT_E_UNSUPPORTED; }
    if ((rc = bind_device(ctx))) return rc;
    smt_corpus *c = new (std::nothrow) smt_corpus();
    if (!c) { set_error("out of host memory"); return SMT_E_NOMEM; }
    c->ctx = ctx; c->dim = D; c->owned = true;
    if (capacity_rows) {
        rc = corpus_reserve(c, capacity_rows);
        if (rc) { delete c; return rc; }
    }
    *out = c;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_corpus_from_device(smt_ctx *ctx, const float *rows_dev, uint64_t n_rows, uint32_t D, smt_corpus **out)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    SMT_REQUIRE(out && (rows_dev || n_rows == 0), "null argument");
    *out = nullptr;
    if (D != SMT_DIM) { set_error("embedding dim %u unsupported", D); return SMT_E_UNSUPPORTED; }
    smt_corpus *c = new (std::nothrow) smt_corpus();
    if (!c) { set_error("out of host memory"); return SMT_E_NOMEM; }
    c->ctx = ctx; c->dim = D; c->owned = false;
    c->d_rows = const_cast<float *>(rows_dev);
    c->rows = n_rows; c->capacity = n_rows;
    // the rows as they are NOW must be in the library's domain (domain.hip); the caller answers for what it writes there later
    if (n_rows) {
        int rc2 = bind_device(ctx);
        if (!rc2) rc2 = require_rows_domain(ctx, rows_dev, n_rows, "smt_corpus_from_device");
        if (rc2) { delete c; return rc2; }
    }
    *out = c;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

void smt_corpus_destroy(smt_corpus *corpus)
{
    if (!corpus) return;
    (void)hipSetDevice(corpus->ctx->device);
    (void)hipStreamSynchronize(corpus->ctx->stream);
    (void)drain_async(corpus->ctx);  // an async select may still be rescoring rows of this corpus
    corpus_writer_destroy(corpus);   // (waits for queued write-ahead jobs)
    if (corpus->owned && corpus->d_rows) (void)hipFree(corpus->d_rows);
    corpus_image_drop(corpus);
    corpus_range_sets_drop(corpus);
    delete corpus;
}

uint64_t smt_corpus_rows(const smt_corpus *corpus) { return corpus ? corpus->rows : 0; }
uint32_t smt_corpus_dim(const smt_corpus *corpus) { return corpus ? corpus->dim : 0; }

int smt_corpus_append_host(smt_corpus *c, const float *rows, uint64_t n_rows, uint64_t *first_row)
try {
    SMT_REQUIRE(c != nullptr, "corpus");
    SMT_REQUIRE(rows || n_rows == 0, "rows");
    int rc = bind_device(c->ctx);
    if (rc) return rc;
    if (first_row) *first_row = c->rows;
    if (n_rows == 0) return SMT_OK;
    if ((rc = corpus_reserve(c, c->rows + n_rows))) return rc;
    SMT_HIP_CHECK(hipMemcpyAsync(c->d_rows + (size_t)c->rows * c->dim, rows, (size_t)n_rows * c->dim * sizeof(float),
                                 hipMemcpyHostToDevice, c->ctx->stream));
    // (checked where they landed, behind the last counted row: a refused batch leaves the corpus as it was)
    if ((rc = require_rows_domain(c->ctx, c->d_rows + (size_t)c->rows * c->dim, n_rows, "smt_corpus_append_host", 0))) return rc;
    c->rows += n_rows;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_corpus_write_rows(smt_corpus *c, uint64_t first_row, const float *rows, uint64_t n_rows)
try {
    SMT_REQUIRE(c != nullptr && (rows || n_rows == 0), "null argument");
    SMT_REQUIRE(first_row + n_rows <= c->rows, "row range outside the corpus");
    int rc = bind_device(c->ctx);
    if (rc) return rc;
    if (n_rows == 0) return SMT_OK;
    {
        const int64_t bad = first_outside_domain_host(rows, n_rows, c->dim);   // (before anything is overwritten)
        if (bad >= 0) {
            set_error("smt_corpus_write_rows: row %lld of the %llu given is outside the library's domain (finite, largest magnitude 0 "
                      "or within [2^-40, 2^40])", (long long)bad, (unsigned long long)n_rows);
            return SMT_E_INVALID;
        }
    }
    corpus_writer_drain(c);   // (a write-ahead job may be reading these rows)
    SMT_HIP_CHECK(hipMemcpyAsync(c->d_rows + (size_t)first_row * c->dim, rows, (size_t)n_rows * c->dim * sizeof(float),
                                 hipMemcpyHostToDevice, c->ctx->stream));
    SMT_HIP_CHECK(hipStreamSynchronize(c->ctx->stream));
    c->image_rows = std::min<uint64_t>(c->image_rows, first_row / 32 * 32);   // the operand image is packed again from this tile on
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_corpus_prepack(smt_corpus *c, int enable)
try {
    SMT_REQUIRE(c != nullptr, "corpus");
    int rc = bind_device(c->ctx);
    if (rc) return rc;
    if (!enable) {
        SMT_HIP_CHECK(hipStreamSynchronize(c->ctx->stream));
        corpus_image_drop(c);
        c->image_mode = -1;
        return SMT_OK;
    }
    c->image_mode = 1;
    c->image_rows = 0;   // "the rows are what they are NOW": an adopted corpus may have changed since the last call
    const void *img;
    const uint32_t *zero;
    if ((rc = corpus_image_sync(c, 0, &img, &zero))) return rc;
    if (c->rows && !img) { set_error("no room for the operand image (%llu bytes)", (unsigned long long)((c->rows + 31) / 32 * 16384)); return SMT_E_NOMEM; }
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

uint64_t smt_corpus_image_bytes(const smt_corpus *c) { return c && c->image ? c->image_cap_tiles * (16384 + 4) : 0; }

int smt_corpus_read_rows(smt_corpus *c, uint64_t first_row, uint64_t n_rows, float *out_host)
try {
    SMT_REQUIRE(c != nullptr && (out_host || n_rows == 0), "null argument");
    SMT_REQUIRE(first_row + n_rows <= c->rows, "row range outside the corpus");
    int rc = bind_device(c->ctx);
    if (rc) return rc;
    if (n_rows == 0) return SMT_OK;
    SMT_HIP_CHECK(hipMemcpyAsync(out_host, c->d_rows + (size_t)first_row * c->dim, (size_t)n_rows * c->dim * sizeof(float),
                                 hipMemcpyDeviceToHost, c->ctx->stream));
    SMT_HIP_CHECK(hipStreamSynchronize(c->ctx->stream));
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_corpus_truncate(smt_corpus *c, uint64_t n_rows)
try {
    SMT_REQUIRE(c != nullptr, "corpus");
    SMT_REQUIRE(n_rows <= c->rows, "cannot truncate to more rows than stored");
    corpus_writer_drain(c);   // (later appends reuse the space: a write-ahead job must not still be reading the old rows there)
    c->rows = n_rows;
    c->image_rows = std::min<uint64_t>(c->image_rows, n_rows / 32 * 32);   // rows appended later land in tiles the image packs again
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

/* ---------------------------------------------------------------- embed ---- */

int smt_embed_device(smt_model *model, const uint32_t *ids_dev, const uint64_t *offsets_dev, uint64_t n_lines,
                     uint32_t max_tokens, float *out_dev)
try {
    SMT_REQUIRE(model != nullptr, "model");
    SMT_REQUIRE(n_lines == 0 || (offsets_dev && out_dev), "null argument");
    int rc = bind_device(model->ctx);
    if (rc) return rc;
    return launch_embed(model->ctx, model->d_table, model->V, model->normalize, ids_dev, offsets_dev, n_lines,
                        max_tokens, out_dev, 0);
} catch (...) { return smt::api_catch(); }

int smt_embed(smt_model *model, const uint32_t *ids, const uint64_t *offsets, uint64_t n_lines, uint32_t max_tokens,
              float *out_host, smt_corpus *append_to, uint64_t *first_row)
try {
    SMT_REQUIRE(model != nullptr, "model");
    SMT_REQUIRE(n_lines == 0 || offsets != nullptr, "offsets");
    smt_ctx *ctx = model->ctx;
    int rc = bind_device(ctx);
    if (rc) return rc;
    if (append_to) {
        SMT_REQUIRE(append_to->ctx == ctx, "corpus belongs to a different context");
        SMT_REQUIRE(append_to->dim == model->D, "corpus dim differs from the model's");
        if (first_row) *first_row = append_to->rows;
    }
    if (n_lines == 0) return SMT_OK;
    SMT_REQUIRE(offsets[0] == 0 || ids != nullptr, "ids");
    const uint64_t n_ids = offsets[n_lines] - offsets[0];
    for (uint64_t i = 0; i < n_lines; ++i) SMT_REQUIRE(offsets[i] <= offsets[i + 1], "offsets must be non-decreasing");
    SMT_REQUIRE(n_ids == 0 || ids != nullptr, "ids");

    // staging: ids + rebased offsets (+ output rows when not appending)
    const size_t ids_bytes = ((size_t)n_ids * sizeof(uint32_t) + 15) & ~(size_t)15;
    const size_t off_bytes = ((size_t)(n_lines + 1) * sizeof(uint64_t) + 15) & ~(size_t)15;
    const size_t out_bytes = append_to ? 0 : (size_t)n_lines * model->D * sizeof(float);
    DeviceTemp tmp;
    {
        hipError_t e = hipMalloc(&tmp.p, ids_bytes + off_bytes + out_bytes + 16);
        if (e != hipSuccess) { set_error("hipMalloc for embed staging: %s", hipGetErrorString(e)); return SMT_E_NOMEM; }
    }
    uint32_t *d_ids = reinterpret_cast<uint32_t *>(tmp.p);
    uint64_t *d_off = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(tmp.p) + ids_bytes);
    float *d_out = reinterpret_cast<float *>(reinterpret_cast<char *>(tmp.p) + ids_bytes + off_bytes);

    std::vector<uint64_t> rebased(n_lines + 1);
    for (uint64_t i = 0; i <= n_lines; ++i) rebased[i] = offsets[i] - offsets[0];
    if (n_ids)
        SMT_HIP_CHECK(hipMemcpyAsync(d_ids, ids + offsets[0], (size_t)n_ids * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    SMT_HIP_CHECK(hipMemcpyAsync(d_off, rebased.data(), (size_t)(n_lines + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));

    if (append_to) {
        if ((rc = corpus_reserve(append_to, append_to->rows + n_lines))) return rc;
        d_out = append_to->d_rows + (size_t)append_to->rows * append_to->dim;
    }
    rc = launch_embed(ctx, model->d_table, model->V, model->normalize, d_ids, d_off, n_lines, max_tokens, d_out,
                      std::max<uint64_t>(n_ids, 1));
    if (rc) return rc;
    if (out_host)
        SMT_HIP_CHECK(hipMemcpyAsync(out_host, d_out, (size_t)n_lines * model->D * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (append_to) {
        // a table with non-finite or absurd entries pools rows no search kernel answers for: they do not enter the corpus
        if ((rc = require_rows_domain(ctx, d_out, n_lines, "smt_embed (rows pooled from the table)", 0))) return rc;
        append_to->rows += n_lines;
    }
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

/* ------------------------------------------------------------------ ids ---- */

uint64_t smt_fnv1a_hash(const uint8_t *bytes, uint64_t n) { return fnv1a(bytes, n); }

uint64_t smt_doc_meta_id(const char *path)
{
    // DocMeta::id, reference src/workspace/store.rs:75-80
    return path ? fnv1a(reinterpret_cast<const uint8_t *>(path), strlen(path)) : fnv1a(nullptr, 0);
}

uint64_t smt_line_embedding_id(const char *path, int32_t line_number)
{
    // LineEmbedding::id, reference src/workspace/store.rs:82-89: path bytes || i32 LE
    std::string b(path ? path : "");
    const uint32_t u = (uint32_t)line_number;
    for (int i = 0; i < 4; ++i) b.push_back((char)((u >> (8 * i)) & 0xff));
    return fnv1a(reinterpret_cast<const uint8_t *>(b.data()), b.size());
}

}  // extern "C"
