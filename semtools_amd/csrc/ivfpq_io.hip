// ivfpq_io.hip -- life cycle of an index object: destroy, info, list sizes, save / load (64-byte header + the device arrays).
#include "ivfpq.h"

using namespace smt;

int smt::ivf_compute_max_list(smt_ivfpq *ix)
{
    std::vector<uint64_t> off(ix->nlist + 1);
    IVF_HIP(hipMemcpy(off.data(), ix->d_offsets, off.size() * 8, hipMemcpyDeviceToHost));
    ix->max_list = 0;
    for (uint32_t l = 0; l < ix->nlist; ++l) ix->max_list = std::max<uint64_t>(ix->max_list, off[l + 1] - off[l]);
    return SMT_OK;
}

extern "C" {

void smt_ivfpq_destroy(smt_ivfpq *ix)
{
    if (!ix) return;
    // The index is documented to be destroyed BEFORE its corpus, but a caller that gets the order wrong (a garbage
    // collector at interpreter exit, an error path) must not turn that into a use-after-free: nothing here touches the
    // corpus or its context -- the device ordinal was recorded at build / load time, and hipDeviceSynchronize covers
    // whatever stream the index's last kernels ran on.
    if (ix->device >= 0) { (void)hipSetDevice(ix->device); (void)hipDeviceSynchronize(); }
    for (void *p : {(void *)ix->d_centroids, (void *)ix->d_cnorm_half, (void *)ix->d_codebooks, (void *)ix->d_codes, (void *)ix->d_ids,
                    (void *)ix->d_offsets, (void *)ix->d_basis, (void *)ix->d_lscale})
        if (p) (void)hipFree(p);
    delete ix;
}

int smt_ivfpq_info(const smt_ivfpq *ix, uint64_t *n_rows, uint32_t *nlist, uint64_t *index_bytes, double *build_ms4)
try {
    SMT_REQUIRE(ix != nullptr, "index");
    if (n_rows) *n_rows = ix->n_rows;
    if (nlist) *nlist = ix->nlist;
    if (index_bytes)
        *index_bytes = (uint64_t)ix->n_rows * (PQ_M + 4) + (uint64_t)ix->nlist * 256 * 4 + (uint64_t)PQ_M * PQ_K * PQ_DSUB * 4 +
                       (uint64_t)(ix->nlist + 1) * 8 +
                       (ix->kind == 1 ? (uint64_t)ix->nlist * LP_DIMS * 257 * 4 : 0);
    if (build_ms4) for (int i = 0; i < 4; ++i) build_ms4[i] = ix->build_ms[i];
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_ivfpq_list_sizes(const smt_ivfpq *ix, uint64_t *sizes_host)
try {
    SMT_REQUIRE(ix && sizes_host, "null argument");
    std::vector<uint64_t> off(ix->nlist + 1);
    IVF_HIP(hipSetDevice(ix->corpus->ctx->device));
    IVF_HIP(hipMemcpy(off.data(), ix->d_offsets, off.size() * 8, hipMemcpyDeviceToHost));
    for (uint32_t l = 0; l < ix->nlist; ++l) sizes_host[l] = off[l + 1] - off[l];
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

// Shared by the host and the device entry points.  queries: host pointer (queries_on_device = false, staged into
// the scratch) or device pointer; the k best (row, exact distance) pairs and the counts are written to the DEVICE
}  // extern "C"

// ---------------------------------------------------------------- persistence
// File = 64-byte little-endian header + the six device arrays in a fixed order.  The index refers to
// rows of a corpus by position, so it is only valid beside the corpus it was built on: load checks
// the row count (the workspace store keeps both files in one directory and rebuilds on mismatch).
namespace {
struct IvfFileHeader {
    char magic[8];      // "SMTIVFP1"
    uint32_t nlist, m, nbits, dim;
    uint64_t n_rows;
    uint8_t pad[32];
};
static_assert(sizeof(IvfFileHeader) == 64, "header layout");

bool write_dev(FILE *f, const void *d, size_t bytes, hipStream_t st, std::vector<char> &buf)
{
    const size_t chunk = (size_t)64 << 20;
    if (buf.size() < std::min(bytes, chunk)) buf.resize(std::min(bytes, chunk));
    for (size_t o = 0; o < bytes; o += chunk) {
        const size_t n = std::min(chunk, bytes - o);
        if (hipMemcpyAsync(buf.data(), (const char *)d + o, n, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
        if (hipStreamSynchronize(st) != hipSuccess) return false;
        if (fwrite(buf.data(), 1, n, f) != n) return false;
    }
    return true;
}

bool read_dev(FILE *f, void *d, size_t bytes, hipStream_t st, std::vector<char> &buf)
{
    const size_t chunk = (size_t)64 << 20;
    if (buf.size() < std::min(bytes, chunk)) buf.resize(std::min(bytes, chunk));
    for (size_t o = 0; o < bytes; o += chunk) {
        const size_t n = std::min(chunk, bytes - o);
        if (fread(buf.data(), 1, n, f) != n) return false;
        if (hipMemcpyAsync((char *)d + o, buf.data(), n, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        if (hipStreamSynchronize(st) != hipSuccess) return false;
    }
    return true;
}
}  // namespace

namespace smt {
__global__ void count_ids_out_of_range_kernel(const uint32_t *ids, uint64_t n, uint32_t n_rows, unsigned int *bad)
{
    unsigned int mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        mine += ids[i] >= n_rows ? 1u : 0u;
    if (mine) atomicAdd(bad, mine);
}
}  // namespace smt

extern "C" {

int smt_ivfpq_save(smt_ivfpq *ix, const char *path)
try {
    SMT_REQUIRE(ix && path, "null argument");
    smt_ctx *ctx = ix->corpus->ctx;
    IVF_HIP(hipSetDevice(ctx->device));
    { int rc_drain = smt::drain_async(ctx); if (rc_drain) return rc_drain; }
    IVF_HIP(hipStreamSynchronize(ctx->stream));
    FILE *f = fopen(path, "wb");
    if (!f) { smt::set_error("cannot open '%s' for writing: %s", path, strerror(errno)); return SMT_E_IO; }
    IvfFileHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "SMTIVFP1", 8);
    h.nlist = ix->nlist;
    h.m = PQ_M;
    h.nbits = 8;
    h.dim = 256;
    h.n_rows = ix->n_rows;
    h.pad[0] = 0;  // (1 marked an index built with the int8 refinement copy, removed in round 3: load refuses such files)
    h.pad[1] = (uint8_t)ix->kind; // 1 = per-list PCA bases + scalar codes follow the codes
    std::vector<char> buf;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    ok = ok && write_dev(f, ix->d_centroids, (size_t)ix->nlist * 256 * 4, ctx->stream, buf);
    ok = ok && write_dev(f, ix->d_cnorm_half, (size_t)ix->nlist * 4, ctx->stream, buf);
    ok = ok && write_dev(f, ix->d_codebooks, (size_t)PQ_M * PQ_K * PQ_DSUB * 4, ctx->stream, buf);
    ok = ok && write_dev(f, ix->d_offsets, (size_t)(ix->nlist + 1) * 8, ctx->stream, buf);
    ok = ok && write_dev(f, ix->d_ids, (size_t)ix->n_rows * 4, ctx->stream, buf);
    ok = ok && write_dev(f, ix->d_codes, (size_t)ix->n_rows * PQ_M, ctx->stream, buf);
    if (ix->kind == 1) {
        ok = ok && write_dev(f, ix->d_basis, (size_t)ix->nlist * LP_DIMS * 256 * 4, ctx->stream, buf);
        ok = ok && write_dev(f, ix->d_lscale, (size_t)ix->nlist * LP_DIMS * 4, ctx->stream, buf);
    }
    if (fclose(f) != 0) ok = false;
    if (!ok) { smt::set_error("short write to '%s'", path); return SMT_E_IO; }
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_ivfpq_load(smt_corpus *corpus, const char *path, smt_ivfpq **out)
try {
    SMT_REQUIRE(corpus && path && out, "null argument");
    *out = nullptr;
    smt_ctx *ctx = corpus->ctx;
    IVF_HIP(hipSetDevice(ctx->device));
    { int rc_drain = smt::drain_async(ctx); if (rc_drain) return rc_drain; }
    FILE *f = fopen(path, "rb");
    if (!f) { smt::set_error("cannot open '%s': %s", path, strerror(errno)); return SMT_E_IO; }
    std::unique_ptr<FILE, int (*)(FILE *)> fguard(f, fclose);
    IvfFileHeader h;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "SMTIVFP1", 8) != 0) {
        smt::set_error("'%s' is not an IVF-PQ index file", path);
        return SMT_E_IO;
    }
    if (h.m != PQ_M || h.nbits != 8 || h.dim != 256 || h.nlist < 32 || h.nlist > PROBE_MAX_LISTS || h.nlist % 32) {
        smt::set_error("'%s': unsupported index geometry (nlist %u, m %u, nbits %u, dim %u)", path, h.nlist, h.m, h.nbits, h.dim);
        return SMT_E_UNSUPPORTED;
    }
    if (h.n_rows > corpus->rows) {  // (fewer is fine: the index covers a prefix, smt_ivfpq_append takes in the rest)
        smt::set_error("'%s' indexes %llu rows but the corpus holds %llu: rebuild", path, (unsigned long long)h.n_rows,
                       (unsigned long long)corpus->rows);
        return SMT_E_INVALID;
    }
    smt_ivfpq *ix = new (std::nothrow) smt_ivfpq();
    if (!ix) { smt::set_error("out of host memory"); return SMT_E_NOMEM; }
    std::unique_ptr<smt_ivfpq, void (*)(smt_ivfpq *)> guard(ix, smt_ivfpq_destroy);
    ix->corpus = corpus;
    ix->device = corpus->ctx->device;
    ix->n_rows = h.n_rows;
    ix->nlist = h.nlist;
    const size_t N = (size_t)h.n_rows;
    IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_centroids), (size_t)h.nlist * 256 * 4));
    IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_cnorm_half), (size_t)h.nlist * 4));
    IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_codebooks), (size_t)PQ_M * PQ_K * PQ_DSUB * 4));
    IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_offsets), (size_t)(h.nlist + 1) * 8));
    IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_ids), std::max<size_t>(N * 4, 16)));
    IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_codes), std::max<size_t>(N * PQ_M, 16)));
    std::vector<char> buf;
    bool ok = read_dev(f, ix->d_centroids, (size_t)h.nlist * 256 * 4, ctx->stream, buf);
    ok = ok && read_dev(f, ix->d_cnorm_half, (size_t)h.nlist * 4, ctx->stream, buf);
    ok = ok && read_dev(f, ix->d_codebooks, (size_t)PQ_M * PQ_K * PQ_DSUB * 4, ctx->stream, buf);
    ok = ok && read_dev(f, ix->d_offsets, (size_t)(h.nlist + 1) * 8, ctx->stream, buf);
    ok = ok && read_dev(f, ix->d_ids, N * 4, ctx->stream, buf);
    ok = ok && read_dev(f, ix->d_codes, N * PQ_M, ctx->stream, buf);
    ix->kind = h.pad[1];
    if (ix->kind > 1) { smt::set_error("'%s': unknown index kind %u", path, ix->kind); return SMT_E_IO; }
    if (ok && ix->kind == 1) {
        IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_basis), (size_t)h.nlist * LP_DIMS * 256 * 4));
        IVF_HIP(hipMalloc(reinterpret_cast<void **>(&ix->d_lscale), (size_t)h.nlist * LP_DIMS * 4));
        ok = ok && read_dev(f, ix->d_basis, (size_t)h.nlist * LP_DIMS * 256 * 4, ctx->stream, buf);
        ok = ok && read_dev(f, ix->d_lscale, (size_t)h.nlist * LP_DIMS * 4, ctx->stream, buf);
    }
    if (!ok) { smt::set_error("'%s' is truncated or unreadable", path); return SMT_E_IO; }
    if (h.pad[0] == 1) { smt::set_error("'%s' was built with the int8 refinement stage (removed): rebuild the index", path); return SMT_E_INVALID; }
    // the list table must be consistent with the row count, or the ADC kernel would read out of bounds
    std::vector<uint64_t> offs((size_t)h.nlist + 1);
    IVF_HIP(hipMemcpy(offs.data(), ix->d_offsets, offs.size() * 8, hipMemcpyDeviceToHost));
    bool sane = offs[0] == 0 && offs[h.nlist] == h.n_rows;
    for (uint32_t l = 0; sane && l < h.nlist; ++l) sane = offs[l] <= offs[l + 1];
    if (!sane) { smt::set_error("'%s': corrupt list offsets", path); return SMT_E_IO; }
    // ... and every stored row id must name a row of THIS corpus: the re-score gathers corpus rows by id (a stale or corrupt file would otherwise read out of bounds)
    if (N > 0) {
        int rc_s = smt::ensure_scratch(ctx, 64);
        if (rc_s) return rc_s;
        unsigned int *d_bad = reinterpret_cast<unsigned int *>(ctx->d_scratch);
        IVF_HIP(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
        hipLaunchKernelGGL(count_ids_out_of_range_kernel, dim3((unsigned)std::min<uint64_t>((N + 255) / 256, 65535)), dim3(256), 0,
                           ctx->stream, ix->d_ids, N, (uint32_t)std::min<uint64_t>(h.n_rows, 0xFFFFFFFFull), d_bad);
        unsigned int bad = 0;
        IVF_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
        IVF_HIP(hipStreamSynchronize(ctx->stream));
        if (bad) { smt::set_error("'%s': %u row ids outside the corpus (stale or corrupt index file)", path, bad); return SMT_E_IO; }
    }
    ix->max_list = 0;
    for (uint32_t l = 0; l < h.nlist; ++l) ix->max_list = std::max<uint64_t>(ix->max_list, offs[l + 1] - offs[l]);
    *out = guard.release();
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

}  // extern "C"