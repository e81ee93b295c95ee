// group.cpp -- the multi-GPU half of libsemtools_hip.so: a GROUP of GPUs behind the C ABI, a corpus row-sharded
// over them, and searches whose only collective is one RCCL all-gather of the per-shard top-k lists.
//
// No reference counterpart: the reference is single-process CPU code (src/bin/semtools.rs:134-135 runs the whole
// search synchronously from one task).  That caller is what this file is shaped for: ONE host thread calls
// smt_init / smt_sharded_search and the library drives every GPU of the node from it --
//   * single process:     smt_group_create(devices, n)  -> ncclCommInitAll, one context + stream per device;
//   * one rank per process (torchrun, MPI): smt_group_unique_id on rank 0, broadcast the 128 bytes by any
//     means, smt_group_create_rank(device, rank, n_ranks, id) -> ncclCommInitRank.  Calls are then SPMD: every
//     process makes the same calls with the same host arguments.
// Partitioning: contiguous row ranges, rows_per_rank = ceil(N / n_ranks), so a document's lines and
// the path-subset ranges of workspace searches stay ranges.  Exchange: rank r's select stage writes its k best
// (global row, exact f64 distance) pairs into one packed buffer; ncclAllGather moves k x 16 B per query per rank
// (latency-bound: xGMI bandwidth is irrelevant); merge_topk_kernel reduces the n_ranks lists to the global
// top-k.  Threshold mode (variable result sizes): all-gather of the counts, then ONE all-gather of a
// max-count-padded buffer, then a host merge of the sorted lists.
//
// RCCL is loaded lazily (dlopen) when the first group is created: single-GPU users -- the `semtools search` CLI
// -- never pay for mapping it.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <stdexcept>
#include <system_error>
#include <thread>

#include <functional>

#include "group.h"

namespace smt {

// ------------------------------------------------------------------ RCCL, loaded on demand
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static RcclApi g_rccl;

static int load_rccl()
{
    if (g_rccl.handle) return SMT_OK;
    // An RCCL already mapped into the process (PyTorch ships its own copy under the same SONAME) wins: two RCCLs
    // in one process would each own a set of IPC handles and proxy threads.
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { set_error("cannot load RCCL (librccl.so.1): %s", dlerror()); return SMT_E_HIP; }
#define SMT_RCCL_SYM(name)                                                                  \
    g_rccl.name = reinterpret_cast<decltype(g_rccl.name)>(dlsym(h, "nccl" #name));          \
    if (!g_rccl.name) { set_error("RCCL lacks nccl" #name); dlclose(h); return SMT_E_HIP; }
    SMT_RCCL_SYM(GetVersion)
    SMT_RCCL_SYM(GetUniqueId)
    SMT_RCCL_SYM(CommInitRank)
    SMT_RCCL_SYM(CommInitAll)
    SMT_RCCL_SYM(CommDestroy)
    SMT_RCCL_SYM(CommCount)
    SMT_RCCL_SYM(CommUserRank)
    SMT_RCCL_SYM(AllGather)
    SMT_RCCL_SYM(AllReduce)
    SMT_RCCL_SYM(GroupStart)
    SMT_RCCL_SYM(GroupEnd)
    SMT_RCCL_SYM(GetErrorString)
#undef SMT_RCCL_SYM
    g_rccl.handle = h;
    return SMT_OK;
}

#define SMT_NCCL_CHECK(expr)                                                                   \
    do {                                                                                       \
        ncclResult_t _r = (expr);                                                              \
        if (_r != ncclSuccess) {                                                               \
            smt::set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
            return SMT_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)

}  // namespace smt

using namespace smt;

namespace smt {

static smt_group *g_default_group = nullptr;

int group_bind(smt_group *g, int i)
{
    SMT_HIP_CHECK(hipSetDevice(g->ctx[i]->device));
    return SMT_OK;
}

int ensure_dev(smt_group *g, int i, size_t bytes)
{
    GroupBuf &b = g->buf[i];
    if (bytes <= b.dev_bytes) return SMT_OK;
    smt_ctx *c = g->ctx[i];
    int rc = drain_async(c);
    if (rc) return rc;
    SMT_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->aux_stream) SMT_HIP_CHECK(hipStreamSynchronize(c->aux_stream));
    if (b.dev) SMT_HIP_CHECK(hipFree(b.dev));
    b.dev = nullptr;
    b.dev_bytes = 0;
    const size_t want = std::max(bytes, (size_t)1 << 16);
    SMT_HIP_CHECK(hipMalloc(&b.dev, want));
    b.dev_bytes = want;
    return SMT_OK;
}

int ensure_host(smt_group *g, int i, size_t bytes)
{
    GroupBuf &b = g->buf[i];
    if (bytes <= b.pinned_bytes) return SMT_OK;
    SMT_HIP_CHECK(hipStreamSynchronize(g->ctx[i]->stream));
    if (b.pinned) SMT_HIP_CHECK(hipHostFree(b.pinned));
    b.pinned = nullptr;
    b.pinned_bytes = 0;
    const size_t want = std::max(bytes, (size_t)1 << 16);
    SMT_HIP_CHECK(hipHostMalloc(&b.pinned, want, hipHostMallocDefault));
    b.pinned_bytes = want;
    return SMT_OK;
}

// All-gather `words` u64 per rank: send_off/recv_off are BYTE offsets into each local device's exchange buffer.
// The streams are the contexts' main streams, or their aux streams where on_aux[i] (async select pipeline).
int allgather_words(smt_group *g, size_t send_off, size_t recv_off, size_t words, const std::vector<char> *on_aux)
{
    if (g->copies) {
        // Copy transport (logical ranks of ONE process): gather into rank 0's receive buffer, then every other rank copies the
        // whole block -- 2 n copies and ~5 n event calls instead of the n x n copies of rounds 2-3 (192 HIP calls per exchange at 8
        // ranks: ~0.45 of the 0.58 ms one host thread needed to issue an 8-shard search, profiles/r04_group_issue.json).
        auto stream_of = [&](int i) { return (on_aux && (*on_aux)[i]) ? g->ctx[i]->aux_stream : g->ctx[i]->stream; };
        const int n = g->n_local;
        for (int j = 0; j < n; ++j) {
            SMT_HIP_CHECK(hipSetDevice(g->ctx[j]->device));
            SMT_HIP_CHECK(hipEventRecord(g->ev_ready[j], stream_of(j)));
        }
        SMT_HIP_CHECK(hipSetDevice(g->ctx[0]->device));
        char *dst0 = reinterpret_cast<char *>(g->buf[0].dev) + recv_off;
        for (int j = 0; j < n; ++j) {
            if (j != 0) SMT_HIP_CHECK(hipStreamWaitEvent(stream_of(0), g->ev_ready[j], 0));
            const char *src = reinterpret_cast<const char *>(g->buf[j].dev) + send_off;
            SMT_HIP_CHECK(hipMemcpyPeerAsync(dst0 + (size_t)j * words * 8, g->ctx[0]->device, src, g->ctx[j]->device, words * 8, stream_of(0)));
        }
        SMT_HIP_CHECK(hipEventRecord(g->ev_done[0], stream_of(0)));       // every send buffer has been read; rank 0 holds the block
        for (int i = 1; i < n; ++i) {
            SMT_HIP_CHECK(hipSetDevice(g->ctx[i]->device));
            SMT_HIP_CHECK(hipStreamWaitEvent(stream_of(i), g->ev_done[0], 0));   // (also: rank i may overwrite its send buffer after this)
            SMT_HIP_CHECK(hipMemcpyPeerAsync(reinterpret_cast<char *>(g->buf[i].dev) + recv_off, g->ctx[i]->device, dst0, g->ctx[0]->device,
                                             (size_t)n * words * 8, stream_of(i)));
            SMT_HIP_CHECK(hipEventRecord(g->ev_done[i], stream_of(i)));
        }
        // rank 0 may not overwrite its receive block (the next exchange) before every rank has copied it
        SMT_HIP_CHECK(hipSetDevice(g->ctx[0]->device));
        for (int i = 1; i < n; ++i) SMT_HIP_CHECK(hipStreamWaitEvent(stream_of(0), g->ev_done[i], 0));
        return SMT_OK;
    }
    SMT_NCCL_CHECK(g_rccl.GroupStart());
    for (int i = 0; i < g->n_local; ++i) {
        (void)hipSetDevice(g->ctx[i]->device);
        char *base = reinterpret_cast<char *>(g->buf[i].dev);
        hipStream_t st = (on_aux && (*on_aux)[i]) ? g->ctx[i]->aux_stream : g->ctx[i]->stream;
        ncclResult_t r = g_rccl.AllGather(base + send_off, base + recv_off, words, ncclUint64, g->comm[i], st);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            set_error("ncclAllGather failed: %s", g_rccl.GetErrorString(r));
            return SMT_E_HIP;
        }
    }
    SMT_NCCL_CHECK(g_rccl.GroupEnd());
    return SMT_OK;
}

// ---------------------------------------------------------------- peer transport (one-process groups)
// Rank j, after the last kernel that writes its list on `st`.
static int peer_publish(smt_group *g, int j, hipStream_t st)
{
    SMT_HIP_CHECK(hipEventRecord(g->ev_ready[j], st));
    g->pub_stream[j] = st;
    return SMT_OK;
}

// Rank j's issuer, right behind peer_publish(j): the merge that local device i will launch on `st_i` waits for rank j's list.
// (The wait names a stream of ANOTHER device when i != j: legal -- a stream carries its device -- and checked by the self-test.)
static int peer_await(smt_group *g, int j, int i, hipStream_t st_i)
{
    if (j == i && g->pub_stream[j] == st_i) return SMT_OK;   // stream order
    SMT_HIP_CHECK(hipStreamWaitEvent(st_i, g->ev_ready[j], 0));
    return SMT_OK;
}

// The caller's thread, after every local rank has published: local device i merges the n_ranks packed lists [nq][2][k_in] that start
// `off` bytes into the ranks' buffers `bases` (exchange buffers, or ring slots), reading them in place, into out_packed
// [nq][2][k_out] on its stream `st`; `done` (may be null) is recorded behind the merge.
static int peer_merge(smt_group *g, int i, hipStream_t st, void *const *bases, size_t off, uint32_t nq, uint32_t k_in, uint32_t k_out,
                      uint64_t *out_packed, hipEvent_t done, bool waits_enqueued = false)
{
    int rc = group_bind(g, i);
    if (rc) return rc;
    MergeSources src;
    for (int j = 0; j < g->n_local; ++j) {
        src.list[j] = reinterpret_cast<const uint64_t *>(reinterpret_cast<const char *>(bases[j]) + off);
        if (waits_enqueued || (j == i && g->pub_stream[j] == st)) continue;   // (the ranks' issuers did it: peer_await / stream order)
        SMT_HIP_CHECK(hipStreamWaitEvent(st, g->ev_ready[j], 0));
    }
    // profiling (smt_prof_enable on device i's context): "exchange" = from this rank's own list being ready to every list being there
    // (the skew between the ranks + what the transport costs), "merge" = the merge kernel
    prof_end_on(g->ctx[i], "exchange", st);
    prof_begin_on(g->ctx[i], "merge", st);
    if ((rc = launch_merge_topk_sources_on(st, src, (uint32_t)g->n_local, nq, k_in, k_out, out_packed))) return rc;
    prof_end_on(g->ctx[i], "merge", st);
    if (done) SMT_HIP_CHECK(hipEventRecord(done, st));
    return SMT_OK;
}

static int sync_every_stream(smt_group *g)
{
    int rc = group_sync_all(g);
    if (rc) return rc;
    for (int i = 0; i < g->n_local; ++i) {
        if ((rc = group_bind(g, i))) return rc;
        if (g->ctx[i]->aux_stream) SMT_HIP_CHECK(hipStreamSynchronize(g->ctx[i]->aux_stream));
    }
    return SMT_OK;
}

static void ring_free(smt_group *g)
{
    smt_group::Ring &r = g->ring;
    for (int i = 0; i < (int)r.dev.size(); ++i)
        if (r.dev[i]) { (void)hipSetDevice(g->ctx[i]->device); (void)hipFree(r.dev[i]); }
    for (auto &row : r.done)
        for (hipEvent_t e : row)
            if (e) (void)hipEventDestroy(e);
    r = smt_group::Ring();
}

// The ring holds slots of at least `slot_bytes`; (re)made -- everything in flight finishes first -- when a call needs larger ones.
static int ring_ensure(smt_group *g, size_t slot_bytes)
{
    smt_group::Ring &r = g->ring;
    if (!r.dev.empty() && slot_bytes <= r.slot_bytes) return SMT_OK;
    int rc = sync_every_stream(g);
    if (rc) return rc;
    ring_free(g);
    r.slot_bytes = (slot_bytes + 255) & ~(size_t)255;
    r.slots = (int)std::min<size_t>(64, std::max<size_t>(2, ((size_t)16 << 20) / r.slot_bytes));
    r.dev.assign(g->n_local, nullptr);
    r.done.assign(r.slots, std::vector<hipEvent_t>(g->n_local, nullptr));
    r.merged.assign(r.slots, 0);
    for (int i = 0; i < g->n_local; ++i) {
        if ((rc = group_bind(g, i))) return rc;
        SMT_HIP_CHECK(hipMalloc(&r.dev[i], (size_t)r.slots * r.slot_bytes));
    }
    return SMT_OK;
}

// The slot of the next exchange, free to be written: the merges that read it `slots` exchanges ago are over (normally long since;
// otherwise the caller's thread waits here -- it may not run more than `slots` searches ahead of the devices).
static int ring_next_slot(smt_group *g, int *slot_out)
{
    smt_group::Ring &r = g->ring;
    const int slot = (int)(r.seq % (uint64_t)r.slots);
    for (int i = 0; r.merged[slot] && i < g->n_local; ++i)
        if (r.merged[slot] >> i & 1) SMT_HIP_CHECK(hipEventSynchronize(r.done[slot][i]));
    r.merged[slot] = 0;
    ++r.seq;
    *slot_out = slot;
    return SMT_OK;
}

static int ring_done_event(smt_group *g, int slot, int i, hipEvent_t *ev)
{
    smt_group::Ring &r = g->ring;
    if (!r.done[slot][i]) {
        // (an event belongs to the device that is current when it is made, and is recorded on a stream of THAT device)
        int rc = group_bind(g, i);
        if (rc) return rc;
        SMT_HIP_CHECK(hipEventCreateWithFlags(&r.done[slot][i], hipEventDisableTiming));
    }
    r.merged[slot] |= (uint64_t)1 << i;
    *ev = r.done[slot][i];
    return SMT_OK;
}

static int group_make_events(smt_group *g)
{
    g->ev_ready.assign(g->n_local, nullptr);
    g->ev_done.assign(g->n_local, nullptr);
    g->pub_stream.assign(g->n_local, nullptr);
    if (const char *w = getenv("SEMTOOLS_GROUP_WAITS")) g->spread_waits = std::string(w) != "caller";
    for (int i = 0; i < g->n_local; ++i) {
        // (the default system-scope release of an event record is what makes a rank's list visible to a reader on another device)
        hipError_t e = hipSetDevice(g->ctx[i]->device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_ready[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_done[i], hipEventDisableTiming);
        if (e != hipSuccess) { set_error("hipEventCreate: %s", hipGetErrorString(e)); return SMT_E_HIP; }
    }
    return SMT_OK;
}

// Every device of a one-process group may read every other's memory?  Enables peer access both ways (RCCL may have done so already).
static bool group_enable_peers(const int *devices, int n)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (i == j) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess || !can) { (void)hipGetLastError(); return false; }
        }
    for (int i = 0; i < n; ++i) {
        if (hipSetDevice(devices[i]) != hipSuccess) { (void)hipGetLastError(); return false; }
        for (int j = 0; j < n; ++j) {
            if (i == j) continue;
            const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
            (void)hipGetLastError();
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return false;
        }
    }
    return true;
}

// The peer transport's self-test (smt_group_create, several devices): every rank's KERNEL writes a small list into its exchange
// buffer, device 0 merges the lists in place, the host checks the merge -- three rounds over the SAME addresses with different
// values, so that a reader serving stale lines from its own cache, or a writer whose lines have not left its L2 when its event
// fires, is caught here and not in an answer.  false = the group falls back to the ncclAllGather transport.
__global__ void peer_test_fill_kernel(uint64_t *list, uint32_t k, uint32_t rank, uint32_t n_ranks, uint32_t round)
{
    const uint32_t i = threadIdx.x;
    if (i >= k) return;
    list[i] = (uint64_t)round * 100000u + rank * 100u + i;                                  // "row"
    const double d = (double)(i * n_ranks + rank) + 0.001 * round;                            // interleaves the ranks' entries
    reinterpret_cast<double *>(list + k)[i] = d;
}

static bool peer_self_test(smt_group *g)
{
    const uint32_t k = 4, n = (uint32_t)g->n_local;
    const size_t out_off = 4096;
    for (int i = 0; i < g->n_local; ++i)
        if (group_bind(g, i) || ensure_dev(g, i, out_off + 1024)) return false;
    std::vector<void *> bases(g->n_local);
    for (int j = 0; j < g->n_local; ++j) bases[j] = g->buf[j].dev;
    for (uint32_t round = 0; round < 3; ++round) {
        for (int j = 0; j < g->n_local; ++j) {
            if (group_bind(g, j)) return false;
            hipLaunchKernelGGL(peer_test_fill_kernel, dim3(1), dim3(64), 0, g->ctx[j]->stream, reinterpret_cast<uint64_t *>(g->buf[j].dev), k,
                               (uint32_t)j, n, round);
            if (hipGetLastError() != hipSuccess || peer_publish(g, j, g->ctx[j]->stream)) return false;
            // (with device j current, as rank j's issuing thread will have it)
            if (g->spread_waits && peer_await(g, j, 0, g->ctx[0]->stream)) {
                (void)hipGetLastError();
                g->spread_waits = false;
                return peer_self_test(g);   // once more from the start, every wait enqueued by the merging device's side
            }
        }
        uint64_t *merged = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(g->buf[0].dev) + out_off);
        if (peer_merge(g, 0, g->ctx[0]->stream, bases.data(), 0, 1, k, k, merged, nullptr, g->spread_waits)) return false;
        uint64_t got[8];
        if (hipMemcpyAsync(got, merged, sizeof(got), hipMemcpyDeviceToHost, g->ctx[0]->stream) != hipSuccess) return false;
        if (group_sync_all(g)) return false;
        for (uint32_t e = 0; e < k; ++e) {   // the k smallest distances are entry 0 of ranks 0 .. k-1 (n >= k), else i * n + rank order
            const uint32_t i = e / n, r = e % n;
            double want_d = (double)(i * n + r) + 0.001 * round, got_d;
            memcpy(&got_d, &got[k + e], 8);
            if (got[e] != (uint64_t)round * 100000u + r * 100u + i || got_d != want_d) return false;
        }
    }
    return true;
}

// $SEMTOOLS_GROUP_TRANSPORT over the default `def`, where the group can use it
static int pick_transport(const smt_group *g, int def)
{
    const char *e = getenv("SEMTOOLS_GROUP_TRANSPORT");
    if (!e || !*e) return def;
    const std::string v(e);
    if (v == "peer" && g->peer_ok) return SMT_TRANSPORT_PEER;
    if (v == "rccl" && !g->copies) return SMT_TRANSPORT_RCCL;
    if (v == "copy" && g->copies) return SMT_TRANSPORT_COPY;
    return def;
}

int group_sync_all(smt_group *g)
{
    for (int i = 0; i < g->n_local; ++i) {
        int rc = group_bind(g, i);
        if (rc) return rc;
        SMT_HIP_CHECK(hipStreamSynchronize(g->ctx[i]->stream));
        if ((rc = drain_async(g->ctx[i]))) return rc;
    }
    return SMT_OK;
}

// A barrier across ranks that also proves the communicator works: all-gather one word per rank.
int group_barrier(smt_group *g)
{
    const size_t words = 1;
    for (int i = 0; i < g->n_local; ++i) {
        int rc = group_bind(g, i);
        if (rc) return rc;
        if ((rc = ensure_dev(g, i, (size_t)(1 + g->n_ranks) * 8 + 64))) return rc;
        const uint64_t me = (uint64_t)(g->first_rank + i);
        SMT_HIP_CHECK(hipMemcpyAsync(g->buf[i].dev, &me, 8, hipMemcpyHostToDevice, g->ctx[i]->stream));
        SMT_HIP_CHECK(hipStreamSynchronize(g->ctx[i]->stream));
    }
    int rc = allgather_words(g, 0, 8, words);
    if (rc) return rc;
    for (int i = 0; i < g->n_local; ++i) {
        if ((rc = group_bind(g, i))) return rc;
        std::vector<uint64_t> got(g->n_ranks);
        SMT_HIP_CHECK(hipMemcpyAsync(got.data(), reinterpret_cast<char *>(g->buf[i].dev) + 8, (size_t)g->n_ranks * 8,
                                     hipMemcpyDeviceToHost, g->ctx[i]->stream));
        SMT_HIP_CHECK(hipStreamSynchronize(g->ctx[i]->stream));
        for (int r = 0; r < g->n_ranks; ++r)
            if (got[r] != (uint64_t)r) { set_error("RCCL all-gather returned rank %llu in slot %d", (unsigned long long)got[r], r); return SMT_E_HIP; }
    }
    return SMT_OK;
}

int group_debug_fail(smt_group *g, int where)
{
    int armed = where;
    if (!g->debug_fail_where.compare_exchange_strong(armed, 0)) return SMT_OK;
    set_error("injected failure (smt_debug_group_fail_next, kind %d) on rank %d", where, g->first_rank);
    return g->debug_fail_code;
}

int group_agree(smt_group *g, int rc)
{
    if (!rc) rc = group_debug_fail(g, SMT_DEBUG_FAIL_AGREE);
    if (g->n_local == g->n_ranks) return rc;   // every rank is in this process: nothing to agree on
    const std::string mine = rc ? smt_last_error() : "";
    for (int i = 0; i < g->n_local; ++i) {
        int rc2 = group_bind(g, i);
        if (!rc2) rc2 = ensure_dev(g, i, (size_t)(1 + g->n_ranks) * 8 + 64);
        if (rc2) return rc2;   // (no way left to tell the others: HIP itself is failing)
        const uint64_t word = (uint64_t)(uint32_t)(rc < 0 ? -rc : rc);
        SMT_HIP_CHECK(hipMemcpyAsync(g->buf[i].dev, &word, 8, hipMemcpyHostToDevice, g->ctx[i]->stream));
        SMT_HIP_CHECK(hipStreamSynchronize(g->ctx[i]->stream));
    }
    int rc2 = allgather_words(g, 0, 8, 1);
    if (rc2) return rc2;
    if ((rc2 = group_bind(g, 0))) return rc2;
    std::vector<uint64_t> got(g->n_ranks);
    SMT_HIP_CHECK(hipMemcpyAsync(got.data(), reinterpret_cast<char *>(g->buf[0].dev) + 8, (size_t)g->n_ranks * 8, hipMemcpyDeviceToHost,
                                 g->ctx[0]->stream));
    if ((rc2 = group_sync_all(g))) return rc2;
    if (rc) { set_error("%s", mine.c_str()); return rc; }
    for (int r = 0; r < g->n_ranks; ++r)
        if (got[r]) { set_error("rank %d failed with status -%llu; this rank gives up with it", r, (unsigned long long)got[r]); return -(int)got[r]; }
    return SMT_OK;
}

// fn(i) for i in [0, n) on n threads that ALL exist before any of them starts: a thread that cannot be created must not leave the
// others waiting for it inside a collective.  false = the threads could not be had and nothing ran.
static bool run_on_threads(int n, const std::function<void(int)> &fn)
{
    std::mutex mu;
    std::condition_variable cv;
    int go = 0;   // 0: wait, 1: run, -1: leave
    std::vector<std::thread> th;
    auto body = [&](int i) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return go != 0; });
        }
        if (go == 1) fn(i);
    };
    bool ok = true;
    try {   // (reserve first: a bad_alloc from the vector's growth, like a system_error from the thread, must find every started thread
            //  still joinable HERE -- unwinding past them would end in std::terminate inside ~thread)
        th.reserve((size_t)n);
        for (int i = 0; i < n; ++i) th.emplace_back(body, i);
    } catch (...) { ok = false; }
    {
        std::lock_guard<std::mutex> lk(mu);
        go = ok ? 1 : -1;
    }
    cv.notify_all();
    for (auto &t : th) t.join();
    if (!ok) set_error("cannot start a host thread per device");
    return ok;
}

static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

static inline long long mono_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// issuing thread t of T: it issues the shares of local ranks t, t + T, ... (T = one per device for a group of real GPUs)
static void worker_main(smt_group *g, int t)
{
    GroupWorkers *w = g->workers;
    const int T = w->n_threads;
    (void)hipSetDevice(g->ctx[t]->device);   // the thread's current device (every entry point binds again: cheap when unchanged)
    uint64_t seen = 0;
    for (;;) {
        // the next call: spin for a while (GroupWorkers), then block
        bool got = false;
        if (w->spin_ns > 0) {
            const long long t0 = mono_ns();
            for (unsigned spins = 1;; ++spins) {
                if (w->epoch.load(std::memory_order_acquire) != seen || w->stop.load(std::memory_order_acquire)) { got = true; break; }
                cpu_relax();
                if ((spins & 127) == 0 && mono_ns() - t0 > w->spin_ns) break;
            }
        }
        if (!got) {
            std::unique_lock<std::mutex> lk(w->mu);
            w->sleepers.fetch_add(1, std::memory_order_seq_cst);
            w->cv_go.wait(lk, [&] { return w->stop.load(std::memory_order_seq_cst) || w->epoch.load(std::memory_order_seq_cst) != seen; });
            w->sleepers.fetch_sub(1, std::memory_order_seq_cst);
        }
        if (w->stop.load(std::memory_order_acquire)) return;
        seen = w->epoch.load(std::memory_order_acquire);
        const std::function<int(int)> *work = w->work;   // (written before the epoch moved)
        for (int i = t; i < g->n_local; i += T) {
            int rc;
            try { rc = (*work)(i); }   // (an exception must reach the caller as a status, not std::terminate the process from this thread)
            catch (const std::bad_alloc &) { set_error("out of host memory"); rc = SMT_E_NOMEM; }
            catch (const std::exception &e) { set_error("%s", e.what()); rc = SMT_E_INVALID; }
            w->rcs[i] = rc;                                  // (slot i is this thread's until `pending` reaches 0)
            if (rc) w->errs[i] = smt_last_error();           // (thread-local: carried back to the caller's thread)
            else w->errs[i].clear();
        }
        if (w->pending.fetch_sub(1, std::memory_order_seq_cst) == 1 && w->caller_blocked.load(std::memory_order_seq_cst)) {
            std::lock_guard<std::mutex> lk(w->mu);
            w->cv_done.notify_one();
        }
    }
}

static void group_stop_workers(smt_group *g);

static int group_start_workers(smt_group *g)
{
    if (g->n_local <= 1 || g->workers) return SMT_OK;
    // SEMTOOLS_GROUP_THREADS=0: the caller's thread issues every device's share itself (A/B, debugging); =1: one thread per local rank;
    // =N >= 2: N issuing threads, each serving every N-th rank.  Default: one per device for a group of real GPUs; FOUR for logical
    // ranks, which share one device, i.e. one runtime lock and one set of hardware queues -- measured on 8 logical shards, host us per
    // search (tools/ab_group_threads.sh, one box): no threads 83, 2 threads 70, 3: 67, 4: 65, 8: 73.  (With the per-rank stream waits of
    // the first peer protocol eight threads were SLOWER than none: 125-170 against 104 us -- the waits serialised on the device's lock.)
    const char *e = getenv("SEMTOOLS_GROUP_THREADS");
    if (e && e[0] == '0') return SMT_OK;
    int n_threads = g->copies ? (g->n_local >= 4 ? 4 : 0) : g->n_local;   // (2 logical ranks: 24 us from the caller's thread, 32 with two threads)
    if (e && atoi(e) >= 2) n_threads = std::min(g->n_local, atoi(e));
    else if (e && e[0] == '1') n_threads = g->n_local;
    if (n_threads < 2) return SMT_OK;
    g->workers = new (std::nothrow) GroupWorkers();
    if (!g->workers) { set_error("out of host memory"); return SMT_E_NOMEM; }
    g->workers->n_threads = n_threads;
    if (const char *sp = getenv("SEMTOOLS_GROUP_SPIN_US")) g->workers->spin_ns = std::max(0LL, atoll(sp)) * 1000;
    g->workers->rcs.assign(g->n_local, SMT_OK);
    g->workers->errs.assign(g->n_local, std::string());
    try {
        g->workers->threads.reserve((size_t)n_threads);
        for (int t = 0; t < n_threads; ++t) g->workers->threads.emplace_back(worker_main, g, t);
    } catch (...) {
        group_stop_workers(g);   // (the ones that exist leave; the group then issues from threads made per call, or from the caller's)
    }
    return SMT_OK;
}

static void group_stop_workers(smt_group *g)
{
    GroupWorkers *w = g->workers;
    if (!w) return;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->stop.store(true, std::memory_order_seq_cst);
    }
    w->cv_go.notify_all();
    for (auto &t : w->threads) t.join();
    delete w;
    g->workers = nullptr;
}

// work(i) for every local device: on the group's issuing threads when it has them (n_local > 1), else on threads made for
// the call (`threads`), else in a row on the caller's thread.  First error wins.
int group_for_each_local(smt_group *g, const std::function<int(int)> &work, bool threads)
{
    std::vector<int> rcs(g->n_local, SMT_OK);
    std::vector<std::string> errs(g->n_local);
    auto run = [&](int i) {
        try { rcs[i] = work(i); }
        catch (const std::bad_alloc &) { set_error("out of host memory"); rcs[i] = SMT_E_NOMEM; }
        catch (const std::exception &e) { set_error("%s", e.what()); rcs[i] = SMT_E_INVALID; }
        if (rcs[i]) errs[i] = smt_last_error();   // (thread-local: carry it back to the caller's thread)
    };
    if (g->n_local == 1 || !threads) {
        for (int i = 0; i < g->n_local; ++i) run(i);
    } else if (GroupWorkers *w = g->workers) {
        // (one caller at a time per group, as for every entry point that takes it)
        w->work = &work;
        w->pending.store(w->n_threads, std::memory_order_relaxed);
        w->epoch.fetch_add(1, std::memory_order_seq_cst);
        if (w->sleepers.load(std::memory_order_seq_cst) > 0) {
            std::lock_guard<std::mutex> lk(w->mu);
            w->cv_go.notify_all();
        }
        bool done = false;
        if (w->spin_ns > 0) {   // the shares of a search are issued in tens of microseconds: wait for them here
            const long long t0 = mono_ns();
            for (unsigned spins = 1;; ++spins) {
                if (w->pending.load(std::memory_order_acquire) == 0) { done = true; break; }
                cpu_relax();
                if ((spins & 127) == 0 && mono_ns() - t0 > 20 * w->spin_ns) break;
            }
        }
        if (!done) {
            std::unique_lock<std::mutex> lk(w->mu);
            w->caller_blocked.store(1, std::memory_order_seq_cst);
            w->cv_done.wait(lk, [&] { return w->pending.load(std::memory_order_seq_cst) == 0; });
            w->caller_blocked.store(0, std::memory_order_seq_cst);
        }
        w->work = nullptr;
        bool any = false;
        for (int i = 0; i < g->n_local; ++i) any = any || w->rcs[i] != SMT_OK;
        if (any) { rcs = w->rcs; errs = w->errs; }
    } else if (!run_on_threads(g->n_local, run)) return SMT_E_NOMEM;
    for (int i = 0; i < g->n_local; ++i)
        if (rcs[i]) {
            if (g->n_ranks > 1) set_error("shard %d: %s", g->first_rank + i, errs[i].c_str());
            else set_error("%s", errs[i].c_str());
            return rcs[i];
        }
    return SMT_OK;
}

static void group_free(smt_group *g)
{
    if (!g) return;
    group_stop_workers(g);
    for (int i = 0; i < (int)g->ctx.size(); ++i) {
        if (!g->ctx[i]) continue;
        (void)hipSetDevice(g->ctx[i]->device);
        (void)hipStreamSynchronize(g->ctx[i]->stream);
        if (g->ctx[i]->aux_stream) (void)hipStreamSynchronize(g->ctx[i]->aux_stream);
    }
    ring_free(g);
    for (int i = 0; i < (int)g->ctx.size(); ++i) {
        if (!g->ctx[i]) continue;
        (void)hipSetDevice(g->ctx[i]->device);
        (void)hipStreamSynchronize(g->ctx[i]->stream);
        (void)drain_async(g->ctx[i]);
        if (i < (int)g->comm.size() && g->comm[i] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g->comm[i]);
        if (i < (int)g->ev_ready.size() && g->ev_ready[i]) (void)hipEventDestroy(g->ev_ready[i]);
        if (i < (int)g->ev_done.size() && g->ev_done[i]) (void)hipEventDestroy(g->ev_done[i]);
        if (i < (int)g->buf.size()) {
            if (g->buf[i].dev) (void)hipFree(g->buf[i].dev);
            if (g->buf[i].pinned) (void)hipHostFree(g->buf[i].pinned);
        }
        if (!g->borrowed) smt_ctx_destroy(g->ctx[i]);
    }
    delete g;
}

static int group_make_contexts(smt_group *g, const int *devices, int n)
{
    g->ctx.assign(n, nullptr);
    g->comm.assign(n, nullptr);
    g->buf.assign(n, GroupBuf());
    g->ar_sums.assign(n, nullptr);
    g->ar_counts.assign(n, nullptr);
    for (int i = 0; i < n; ++i) {
        int rc = smt_ctx_create(devices[i], &g->ctx[i]);
        if (rc) return rc;
    }
    return SMT_OK;
}

static int validate_global_ranges(const smt_range *ranges, uint32_t n, uint64_t rows)
{
    uint64_t prev_end = 0;
    for (uint32_t i = 0; i < n; ++i) {
        SMT_REQUIRE(ranges[i].begin <= ranges[i].end, "range begin > end");
        SMT_REQUIRE(ranges[i].end <= rows, "range extends past the corpus");
        SMT_REQUIRE(i == 0 || ranges[i].begin >= prev_end, "ranges must be sorted and disjoint");
        prev_end = ranges[i].end;
    }
    return SMT_OK;
}

// ---------------------------------------------------------------- generic (host-list) exchange
// Every local shard holds per-query hit lists of any length, sorted (distance asc, row asc), rows global.  All-gather
// of the counts, one all-gather of a max-count-padded buffer, host merge.  `keep` truncates after the merge
// (UINT64_MAX = keep all: search_documents with a threshold, src/search/mod.rs:115-116).
// One query's lists, each sorted by (distance, row), merged into the first `keep` entries of their union in that order: a
// cursor per list, the smallest head taken each time (R is the number of ranks: a linear scan of the heads beats a heap).
// (Round 5: this was std::sort over the union -- 3-4 ms per query with 65 k hits under a threshold.)
struct HitSpan { const uint64_t *rows; const double *dist; uint64_t n; };
static void merge_hit_spans(std::vector<HitSpan> &src, uint64_t keep, LocalHits &out)
{
    uint64_t total = 0;
    for (const HitSpan &sp : src) total += sp.n;
    const uint64_t n_out = std::min<uint64_t>(total, keep);
    out.rows.resize(n_out);
    out.dist.resize(n_out);
    size_t live = 0;
    for (size_t i = 0; i < src.size(); ++i)
        if (src[i].n) src[live++] = src[i];
    src.resize(live);
    for (uint64_t e = 0; e < n_out; ++e) {
        size_t best = 0;
        for (size_t i = 1; i < src.size(); ++i)
            if (src[i].dist[0] < src[best].dist[0] || (src[i].dist[0] == src[best].dist[0] && src[i].rows[0] < src[best].rows[0])) best = i;
        out.rows[e] = src[best].rows[0];
        out.dist[e] = src[best].dist[0];
        ++src[best].rows; ++src[best].dist;
        if (--src[best].n == 0) { src[best] = src.back(); src.pop_back(); }
    }
}

static int exchange_host_lists(smt_group *g, const std::vector<std::vector<LocalHits>> &local /* [n_local][nq] */, uint32_t nq,
                               uint64_t keep, std::vector<LocalHits> &merged)
{
    merged.assign(nq, LocalHits());
    if (nq == 0) return SMT_OK;
    const int R = g->n_ranks;
    int rc;
    std::vector<HitSpan> src;
    // every rank lives in this process (one-process and logical groups): the lists are all here already -- nothing travels
    // ($SEMTOOLS_GROUP_HOST_LISTS=exchange sends them through the devices anyway: the tests' way to run the multi-process path below
    // on the logical ranks of one GPU)
    const char *force = getenv("SEMTOOLS_GROUP_HOST_LISTS");
    if (g->n_local == g->n_ranks && !(force && std::string(force) == "exchange")) {
        for (uint32_t q = 0; q < nq; ++q) {
            src.clear();
            for (int i = 0; i < g->n_local; ++i) src.push_back({local[i][q].rows.data(), local[i][q].dist.data(), local[i][q].rows.size()});
            merge_hit_spans(src, keep, merged[q]);
        }
        return SMT_OK;
    }
    // ---- counts
    for (int i = 0; i < g->n_local; ++i) {
        if ((rc = group_bind(g, i))) return rc;
        if ((rc = ensure_dev(g, i, (size_t)(1 + R) * nq * 8 + 64))) return rc;
        std::vector<uint64_t> cnt(nq);
        for (uint32_t q = 0; q < nq; ++q) cnt[q] = std::min<uint64_t>(local[i][q].rows.size(), keep);   // (nobody needs more than `keep` of a list)
        SMT_HIP_CHECK(hipMemcpyAsync(g->buf[i].dev, cnt.data(), (size_t)nq * 8, hipMemcpyHostToDevice, g->ctx[i]->stream));
        SMT_HIP_CHECK(hipStreamSynchronize(g->ctx[i]->stream));
    }
    if ((rc = allgather_words(g, 0, (size_t)nq * 8, nq))) return rc;
    std::vector<uint64_t> counts((size_t)R * nq);
    if ((rc = group_bind(g, 0))) return rc;
    SMT_HIP_CHECK(hipMemcpyAsync(counts.data(), reinterpret_cast<char *>(g->buf[0].dev) + (size_t)nq * 8, counts.size() * 8,
                                 hipMemcpyDeviceToHost, g->ctx[0]->stream));
    if ((rc = group_sync_all(g))) return rc;
    std::vector<uint64_t> width(nq, 0), off(nq + 1, 0);
    for (uint32_t q = 0; q < nq; ++q) {
        for (int r = 0; r < R; ++r) width[q] = std::max(width[q], counts[(size_t)r * nq + q]);
        off[q + 1] = off[q] + 2 * width[q];
    }
    const size_t words = off[nq];  // per rank
    if (words == 0) return SMT_OK;
    // ---- padded payload: per query [rows | distance bits], width[q] each
    const size_t send_bytes = words * 8, recv_off = (send_bytes + 255) & ~(size_t)255;
    for (int i = 0; i < g->n_local; ++i) {
        if ((rc = group_bind(g, i))) return rc;
        if ((rc = ensure_dev(g, i, recv_off + (size_t)R * send_bytes + 64))) return rc;
        if ((rc = ensure_host(g, i, std::max(send_bytes, i == 0 ? (size_t)R * send_bytes : (size_t)0)))) return rc;
        uint64_t *h = reinterpret_cast<uint64_t *>(g->buf[i].pinned);
        for (uint32_t q = 0; q < nq; ++q) {
            const LocalHits &l = local[i][q];
            const uint64_t mine = std::min<uint64_t>(l.rows.size(), keep);
            uint64_t *rows = h + off[q], *bits = rows + width[q];
            for (uint64_t e = 0; e < width[q]; ++e) {
                if (e < mine) { rows[e] = l.rows[e]; memcpy(bits + e, &l.dist[e], 8); }
                else { rows[e] = UINT64_MAX; bits[e] = 0x7FF0000000000000ull; }
            }
        }
        SMT_HIP_CHECK(hipMemcpyAsync(g->buf[i].dev, h, send_bytes, hipMemcpyHostToDevice, g->ctx[i]->stream));
    }
    if ((rc = allgather_words(g, 0, recv_off, words))) return rc;
    if ((rc = group_bind(g, 0))) return rc;
    uint64_t *all = reinterpret_cast<uint64_t *>(g->buf[0].pinned);
    SMT_HIP_CHECK(hipMemcpyAsync(all, reinterpret_cast<char *>(g->buf[0].dev) + recv_off, (size_t)R * send_bytes,
                                 hipMemcpyDeviceToHost, g->ctx[0]->stream));
    if ((rc = group_sync_all(g))) return rc;
    // ---- merge.  Shards are ascending contiguous row ranges (or pieces dealt in insertion order) and every list is (distance, row)-
    // sorted, so the (distance, row) merge of the lists reproduces the reference's stable sort over the whole corpus (mod.rs:107-111).
    for (uint32_t q = 0; q < nq; ++q) {
        src.clear();
        for (int r = 0; r < R; ++r) {
            const uint64_t *rows = all + (size_t)r * words + off[q];
            src.push_back({rows, reinterpret_cast<const double *>(rows + width[q]), counts[(size_t)r * nq + q]});
        }
        merge_hit_spans(src, keep, merged[q]);
    }
    return SMT_OK;
}

// Per-shard host search of a query subset on every local device (one host thread per device when there are
// several: the K4 / large-k paths synchronise internally and would otherwise serialise the GPUs).
static int local_host_search(smt_sharded_corpus *sc, const float *queries, uint32_t nq, uint32_t top_k, double max_distance,
                             int mode, const smt_range *ranges, uint32_t n_ranges, std::vector<std::vector<LocalHits>> &local)
{
    smt_group *g = sc->group;
    local.assign(g->n_local, std::vector<LocalHits>(nq));
    std::vector<int> rcs(g->n_local, SMT_OK);
    std::vector<std::string> errs(g->n_local);
    auto work = [&](int i) {
        const int r = g->first_rank + i;
        std::vector<smt_range> lr;
        if (n_ranges) {
            layout_localize(sc, r, ranges, n_ranges, lr);
            if (lr.empty()) return;  // the filter leaves this shard nothing
        }
        // a shard cut as ONE range returns global rows by adding its base; a shard of several pieces returns local rows
        // (already in global order: pieces ascend in both numberings) which are mapped piece by piece
        rcs[i] = search_local_host(sc->shard[i], queries, nq, top_k, max_distance, mode, lr.empty() ? nullptr : lr.data(),
                                   (uint32_t)lr.size(), sc->contiguous ? sc->rank_base[r] : 0, local[i]);
        if (rcs[i]) { errs[i] = smt_last_error(); return; }
        if (!sc->contiguous)
            for (LocalHits &h : local[i])
                for (uint64_t &row : h.rows) row = layout_to_global(sc, r, row);
    };
    if (g->n_local == 1) work(0);
    else if (!run_on_threads(g->n_local, work)) return SMT_E_NOMEM;
    for (int i = 0; i < g->n_local; ++i)
        if (rcs[i]) { set_error("shard %d: %s", g->first_rank + i, errs[i].c_str()); return rcs[i]; }
    return SMT_OK;
}

// ---------------------------------------------------------------- all-reduce for shared-centroid IVF builds
__global__ void sum_ranks_i64_kernel(const long long *const *ptrs, int n_ranks, size_t n, long long *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long acc = 0;
    for (int r = 0; r < n_ranks; ++r) acc += ptrs[r][i];
    out[i] = acc;
}
__global__ void sum_ranks_u32_kernel(const unsigned int *const *ptrs, int n_ranks, size_t n, unsigned int *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned int acc = 0;
    for (int r = 0; r < n_ranks; ++r) acc += ptrs[r][i];
    out[i] = acc;
}

static void thread_barrier(smt_group *g)
{
    std::unique_lock<std::mutex> lk(g->ar_mu);
    const uint64_t gen = g->ar_generation;
    if (++g->ar_waiting == g->n_local) {
        g->ar_waiting = 0;
        ++g->ar_generation;
        g->ar_cv.notify_all();
    } else {
        g->ar_cv.wait(lk, [&] { return g->ar_generation != gen; });
    }
}

struct ShareCtx {
    smt_group *g;
    int local;
};

// IvfBuildShare::allreduce for one local rank (called from that rank's host thread)
static int group_allreduce_sums(void *user, long long *sums, size_t n_sums, unsigned int *counts, size_t n_counts)
{
    ShareCtx *sc = static_cast<ShareCtx *>(user);
    smt_group *g = sc->g;
    const int i = sc->local;
    smt_ctx *c = g->ctx[i];
    if (!g->copies) {
        SMT_NCCL_CHECK(g_rccl.AllReduce(sums, sums, n_sums, ncclInt64, ncclSum, g->comm[i], c->stream));
        SMT_NCCL_CHECK(g_rccl.AllReduce(counts, counts, n_counts, ncclUint32, ncclSum, g->comm[i], c->stream));
        return SMT_OK;
    }
    // copy transport: every rank lives in this process (one thread each): meet, sum everybody's buffer, meet, copy back.
    // A rank whose HIP calls fail still passes BOTH barriers (its siblings would wait for it forever) and then reports.
    const size_t b_sums = ((n_sums * 8 + 255) & ~(size_t)255), b_cnt = ((n_counts * 4 + 255) & ~(size_t)255);
    const size_t b_ptr = (((size_t)g->n_local * 16 + 255) & ~(size_t)255);
    int rc = ensure_dev(g, i, b_sums + b_cnt + b_ptr + 64);
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !rc) { set_error("%s: %s", what, hipGetErrorString(e)); rc = SMT_E_HIP; }
    };
    hip_ok(hipStreamSynchronize(c->stream), "all-reduce (sync)");
    {
        std::lock_guard<std::mutex> lk(g->ar_mu);
        g->ar_sums[i] = sums;
        g->ar_counts[i] = counts;
    }
    thread_barrier(g);
    long long *t_sums = nullptr;
    unsigned int *t_cnt = nullptr;
    if (!rc) {
        char *base = reinterpret_cast<char *>(g->buf[i].dev);
        t_sums = reinterpret_cast<long long *>(base);
        t_cnt = reinterpret_cast<unsigned int *>(base + b_sums);
        const long long **d_ps = reinterpret_cast<const long long **>(base + b_sums + b_cnt);
        const unsigned int **d_pc = reinterpret_cast<const unsigned int **>(base + b_sums + b_cnt + (size_t)g->n_local * 8);
        hip_ok(hipMemcpyAsync(d_ps, g->ar_sums.data(), (size_t)g->n_local * 8, hipMemcpyHostToDevice, c->stream), "all-reduce (pointers)");
        hip_ok(hipMemcpyAsync(d_pc, g->ar_counts.data(), (size_t)g->n_local * 8, hipMemcpyHostToDevice, c->stream), "all-reduce (pointers)");
        if (!rc) {
            hipLaunchKernelGGL(sum_ranks_i64_kernel, dim3((unsigned)((n_sums + 255) / 256)), dim3(256), 0, c->stream, d_ps, g->n_local, n_sums, t_sums);
            hipLaunchKernelGGL(sum_ranks_u32_kernel, dim3((unsigned)((n_counts + 255) / 256)), dim3(256), 0, c->stream, d_pc, g->n_local, n_counts, t_cnt);
            hip_ok(hipGetLastError(), "all-reduce (sum kernels)");
        }
        hip_ok(hipStreamSynchronize(c->stream), "all-reduce (sum)");
    }
    thread_barrier(g);   // nobody overwrites its buffer before everybody has read it
    if (rc) return rc;
    SMT_HIP_CHECK(hipMemcpyAsync(sums, t_sums, n_sums * 8, hipMemcpyDeviceToDevice, c->stream));
    SMT_HIP_CHECK(hipMemcpyAsync(counts, t_cnt, n_counts * 4, hipMemcpyDeviceToDevice, c->stream));
    return SMT_OK;
}

// IvfBuildShare::agree: the ranks meet and share a status before the first collective of a build, so that a rank whose
// set-up failed (out of memory ...) takes the others with it instead of leaving them in the all-reduce.
static int group_share_agree(void *user, int rc)
{
    ShareCtx *sc = static_cast<ShareCtx *>(user);
    smt_group *g = sc->g;
    if (!rc && sc->local == 0) rc = group_debug_fail(g, SMT_DEBUG_FAIL_BUILD);
    if (g->n_local == 1) return group_agree(g, rc);
    const std::string mine = rc ? smt_last_error() : "";
    {
        std::lock_guard<std::mutex> lk(g->ar_mu);
        if (rc && !g->ar_failed) g->ar_failed = rc;
    }
    thread_barrier(g);
    const int all = g->ar_failed;
    thread_barrier(g);             // everybody has read the verdict ...
    if (sc->local == 0) g->ar_failed = 0;   // ... before it is cleared for the next build
    thread_barrier(g);
    if (rc) { set_error("%s", mine.c_str()); return rc; }
    if (all) { set_error("another shard of the group failed to set up its index build"); return all; }
    return SMT_OK;
}

}  // namespace smt

extern "C" {

/* ------------------------------------------------------------------ group ---- */

int smt_group_create(const int *devices, int n_dev, smt_group **out)
try {
    SMT_REQUIRE(out != nullptr, "out");
    *out = nullptr;
    SMT_REQUIRE(devices != nullptr && n_dev >= 1, "device list");
    for (int i = 0; i < n_dev; ++i)
        for (int j = 0; j < i; ++j) SMT_REQUIRE(devices[i] != devices[j], "a device may appear once in a group");
    int rc = load_rccl();
    if (rc) return rc;
    smt_group *g = new (std::nothrow) smt_group();
    if (!g) { set_error("out of host memory"); return SMT_E_NOMEM; }
    g->n_ranks = g->n_local = n_dev;
    g->first_rank = 0;
    if ((rc = group_make_contexts(g, devices, n_dev))) { group_free(g); return rc; }
    (void)g_rccl.GetVersion(&g->rccl_version);
    ncclResult_t r = g_rccl.CommInitAll(g->comm.data(), n_dev, devices);
    if (r != ncclSuccess) { set_error("ncclCommInitAll(%d devices): %s", n_dev, g_rccl.GetErrorString(r)); group_free(g); return SMT_E_HIP; }
    if ((rc = group_barrier(g))) { group_free(g); return rc; }  // channel set-up happens on the first collective: do it now
    // the k-lists of a top-k search are read in place by the merging device when every device can map every other's memory
    // (one hive of xGMI-linked GPUs: always, in practice); otherwise they travel through ncclAllGather
    if ((rc = group_make_events(g))) { group_free(g); return rc; }
    g->peer_ok = n_dev <= SMT_MAX_MERGE_SOURCES && (n_dev == 1 || group_enable_peers(devices, n_dev));
    if (g->peer_ok && n_dev > 1 && !peer_self_test(g)) {
        // (never seen; a node whose devices map each other's memory but do not serve it coherently to a reader's kernel)
        fprintf(stderr, "libsemtools_hip: peer reads between the devices of this group failed their self-test: using ncclAllGather\n");
        (void)hipGetLastError();
        g->peer_ok = false;
    }
    g->transport = pick_transport(g, g->peer_ok ? SMT_TRANSPORT_PEER : SMT_TRANSPORT_RCCL);
    if ((rc = group_start_workers(g))) { group_free(g); return rc; }
    *out = g;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_group_create_logical(int device, int n_shards, smt_group **out)
try {
    SMT_REQUIRE(out != nullptr, "out");
    *out = nullptr;
    SMT_REQUIRE(n_shards >= 1 && n_shards <= 64, "1..64 logical shards");
    smt_group *g = new (std::nothrow) smt_group();
    if (!g) { set_error("out of host memory"); return SMT_E_NOMEM; }
    g->n_ranks = g->n_local = n_shards;
    g->first_rank = 0;
    g->copies = true;
    std::vector<int> devs(n_shards, device);
    int rc = group_make_contexts(g, devs.data(), n_shards);
    if (rc) { group_free(g); return rc; }
    if ((rc = group_make_events(g))) { group_free(g); return rc; }
    g->peer_ok = true;   // one device
    if (n_shards > 1 && !peer_self_test(g)) { set_error("the peer-read self-test failed on logical ranks of one device"); group_free(g); return SMT_E_HIP; }
    g->transport = pick_transport(g, SMT_TRANSPORT_PEER);
    if ((rc = group_barrier(g))) { group_free(g); return rc; }
    if ((rc = group_start_workers(g))) { group_free(g); return rc; }
    *out = g;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_group_from_ctx(smt_ctx *ctx, smt_group **out)
try {
    SMT_REQUIRE(ctx && out, "null argument");
    *out = nullptr;
    smt_group *g = new (std::nothrow) smt_group();
    if (!g) { set_error("out of host memory"); return SMT_E_NOMEM; }
    g->n_ranks = g->n_local = 1;
    g->first_rank = 0;
    g->copies = true;      // no communicator: a one-rank group never exchanges anything
    g->borrowed = true;    // the caller keeps (and later destroys) the context
    g->ctx.assign(1, ctx);
    g->comm.assign(1, nullptr);
    g->buf.assign(1, GroupBuf());
    g->ar_sums.assign(1, nullptr);
    g->ar_counts.assign(1, nullptr);
    g->ev_ready.assign(1, nullptr);
    g->ev_done.assign(1, nullptr);
    g->pub_stream.assign(1, nullptr);
    g->transport = SMT_TRANSPORT_COPY;
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_ready[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_done[0], hipEventDisableTiming);
    if (e != hipSuccess) { set_error("hipEventCreate: %s", hipGetErrorString(e)); group_free(g); return SMT_E_HIP; }
    *out = g;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_group_unique_id(void *id_out)
try {
    SMT_REQUIRE(id_out != nullptr, "id_out");
    int rc = load_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == SMT_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    SMT_NCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_group_create_rank(int device, int rank, int n_ranks, const void *unique_id, smt_group **out)
try {
    SMT_REQUIRE(out != nullptr, "out");
    *out = nullptr;
    SMT_REQUIRE(unique_id != nullptr, "unique_id");
    SMT_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "rank / n_ranks");
    int rc = load_rccl();
    if (rc) return rc;
    smt_group *g = new (std::nothrow) smt_group();
    if (!g) { set_error("out of host memory"); return SMT_E_NOMEM; }
    g->n_ranks = n_ranks;
    g->n_local = 1;
    g->first_rank = rank;
    if ((rc = group_make_contexts(g, &device, 1))) { group_free(g); return rc; }
    (void)g_rccl.GetVersion(&g->rccl_version);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    (void)hipSetDevice(device);
    ncclResult_t r = g_rccl.CommInitRank(&g->comm[0], n_ranks, id, rank);
    if (r != ncclSuccess) { set_error("ncclCommInitRank(rank %d of %d): %s", rank, n_ranks, g_rccl.GetErrorString(r)); group_free(g); return SMT_E_HIP; }
    if ((rc = group_barrier(g))) { group_free(g); return rc; }
    *out = g;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

void smt_group_destroy(smt_group *group)
{
    if (group == g_default_group) g_default_group = nullptr;
    group_free(group);
}

int smt_group_info(const smt_group *group, int *n_ranks, int *n_local, int *first_rank, int *rccl_ranks, int *rccl_version)
try {
    SMT_REQUIRE(group != nullptr, "group");
    if (n_ranks) *n_ranks = group->n_ranks;
    if (n_local) *n_local = group->n_local;
    if (first_rank) *first_rank = group->first_rank;
    if (rccl_version) *rccl_version = group->rccl_version;
    if (rccl_ranks) {
        int count = 0;  // copy transport: no communicator
        if (!group->copies) SMT_NCCL_CHECK(g_rccl.CommCount(group->comm[0], &count));  // what the communicator itself reports
        *rccl_ranks = count;
    }
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

smt_ctx *smt_group_ctx(smt_group *group, int local_index)
{
    if (!group || local_index < 0 || local_index >= group->n_local) return nullptr;
    return group->ctx[local_index];
}

int smt_group_synchronize(smt_group *group)
try {
    SMT_REQUIRE(group != nullptr, "group");
    return group_sync_all(group);
} catch (...) { return smt::api_catch(); }

int smt_group_barrier(smt_group *group)
try {
    SMT_REQUIRE(group != nullptr, "group");
    int rc = group_sync_all(group);
    if (rc) return rc;
    return group_barrier(group);
} catch (...) { return smt::api_catch(); }

int smt_group_set_transport(smt_group *group, int transport)
try {
    SMT_REQUIRE(group != nullptr, "group");
    SMT_REQUIRE(transport == SMT_TRANSPORT_RCCL || transport == SMT_TRANSPORT_COPY || transport == SMT_TRANSPORT_PEER, "transport");
    SMT_REQUIRE(transport != SMT_TRANSPORT_PEER || group->peer_ok, "peer transport needs a one-process group whose devices can read each other's memory");
    SMT_REQUIRE(transport != SMT_TRANSPORT_RCCL || !group->copies, "this group has no RCCL communicator");
    SMT_REQUIRE(transport != SMT_TRANSPORT_COPY || group->copies, "copy transport is the all-gather of logical groups");
    int rc = sync_every_stream(group);   // nothing of an earlier exchange is in flight when the protocol changes
    if (rc) return rc;
    group->transport = transport;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_debug_group_fail_next(smt_group *group, int where, int code)
try {
    SMT_REQUIRE(group != nullptr, "group");
    SMT_REQUIRE(where == 0 || where == SMT_DEBUG_FAIL_STAGE || where == SMT_DEBUG_FAIL_AGREE || where == SMT_DEBUG_FAIL_BUILD, "where");
    SMT_REQUIRE(where == 0 || code < 0, "code must be an SMT_E_* status");
    group->debug_fail_code = code;
    group->debug_fail_where.store(where);
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_group_transport(const smt_group *group)
try {
    SMT_REQUIRE(group != nullptr, "group");
    return group->transport;
} catch (...) { return smt::api_catch(); }

int smt_init(const int *devices, int n_dev)
try {
    if (g_default_group) {
        // idempotent for the same device list
        bool same = g_default_group->n_local == n_dev;
        for (int i = 0; same && i < n_dev; ++i) same = devices && g_default_group->ctx[i]->device == devices[i];
        if (same) return SMT_OK;
        set_error("smt_init was already called with a different device list (call smt_shutdown first)");
        return SMT_E_INVALID;
    }
    std::vector<int> all;
    if (!devices || n_dev <= 0) {  // NULL / 0 = every visible GPU
        const int n = smt_device_count();
        if (n <= 0) { if (n == 0) set_error("no HIP device visible: libsemtools_hip has no CPU fallback"); return SMT_E_HIP; }
        for (int i = 0; i < n; ++i) all.push_back(i);
        devices = all.data();
        n_dev = n;
    }
    return smt_group_create(devices, n_dev, &g_default_group);
} catch (...) { return smt::api_catch(); }

int smt_shutdown(void)
try {
    smt_group *g = g_default_group;
    g_default_group = nullptr;
    group_free(g);
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

smt_group *smt_default_group(void) { return g_default_group; }

/* ----------------------------------------------------------------- search ---- */

int smt_sharded_search(smt_sharded_corpus *sc, const float *queries, uint32_t nq, uint32_t top_k, double max_distance, int mode,
                       const smt_range *ranges, uint32_t n_ranges, uint64_t *out_rows, double *out_dist, uint64_t *out_counts,
                       uint64_t out_cap)
try {
    SMT_REQUIRE(sc != nullptr, "corpus");
    SMT_REQUIRE(mode == SMT_MODE_DOCUMENTS || mode == SMT_MODE_WORKSPACE, "mode");
    SMT_REQUIRE(nq == 0 || (queries && out_counts), "null argument");
    SMT_REQUIRE(n_ranges == 0 || ranges != nullptr, "ranges");
    smt_group *g = sc->group;
    if (nq == 0) return SMT_OK;
    // (domain.hip; SPMD callers pass the same host arguments, so every rank refuses together, before any collective)
    if (int rcq = require_queries_domain_host(queries, nq, "smt_sharded_search")) return rcq;
    if (g->n_ranks == 1)   // one shard: the single-GPU call, nothing to exchange
        return smt_search(sc->shard[0], queries, nq, top_k, max_distance, mode, ranges, n_ranges, 0, out_rows, out_dist, out_counts, out_cap);
    for (uint32_t q = 0; q < nq; ++q) out_counts[q] = 0;
    const uint64_t total = sc->total();
    int rc;
    if (n_ranges && (rc = validate_global_ranges(ranges, n_ranges, total))) return rc;
    uint64_t n_virtual = total;
    if (n_ranges) {
        n_virtual = 0;
        for (uint32_t i = 0; i < n_ranges; ++i) n_virtual += ranges[i].end - ranges[i].begin;
    }
    const bool has_thr = !std::isnan(max_distance);
    const bool all_under_threshold = (mode == SMT_MODE_DOCUMENTS) && has_thr;
    if (n_virtual == 0) return SMT_OK;
    if (!all_under_threshold && top_k == 0) return SMT_OK;

    std::vector<LocalHits> hits(nq);
    const uint32_t K = (uint32_t)std::min<uint64_t>(top_k, n_virtual);
    const bool device_exchange = !all_under_threshold && K <= SCAN_MAX_K && (uint64_t)g->n_ranks * K <= 8192;
    std::vector<uint32_t> redo;  // queries answered through the host-list exchange
    if (device_exchange) {
        // ---- per-shard scan + select -> packed lists -> ONE all-gather -> device merge
        // per rank: the lists, one "uncertain" word per query, one STATUS word -- a rank whose local stage failed still
        // takes part in the all-gather and says so there, so that every process returns the error together (a rank
        // that simply left would strand the others inside the collective)
        const size_t list_words = (size_t)nq * 2 * K, rank_words = list_words + nq + 1;
        const size_t q_bytes = ((size_t)nq * SMT_DIM * 4 + 255) & ~(size_t)255;
        const size_t loc_off = q_bytes;
        const size_t gath_off = loc_off + ((rank_words * 8 + 255) & ~(size_t)255);
        const size_t out_off = gath_off + (((size_t)g->n_ranks * rank_words * 8 + 255) & ~(size_t)255);
        const size_t dev_bytes = out_off + list_words * 8 + 64;
        const int ws = (mode == SMT_MODE_WORKSPACE && has_thr) ? 1 : 0;
        const float thr_score = 1.0f - (float)max_distance;  // store.rs:502-503
        int local_rc = SMT_OK;
        std::string local_err;
        // Pinned staging per local device: [queries][status word] (+ device 0: the merged lists and every rank's flags).  A device's
        // share only ENQUEUES -- upload, scan, select, status -- and the shares of a one-process group are issued at once by the
        // group's issuing threads (group_for_each_local), so the shards scan concurrently and the caller's thread pays for one.
        // (a stack status word once forced a stream sync per device here, i.e. the SUM of the shard times.)
        const size_t flag_words = (size_t)g->n_ranks * (nq + 1);
        const size_t pin_status_off = q_bytes, pin_res_off = q_bytes + 64;
        // peer transport (one-process groups): no gather -- device 0's merge reads the ranks' lists in place, every rank copies
        // its own flag words to its own pinned block on its own stream (issued by its thread)
        const bool peer = g->transport == SMT_TRANSPORT_PEER;
        std::vector<int> stage_rcs(g->n_local, SMT_OK);
        std::vector<std::string> stage_errs(g->n_local);
        rc = group_for_each_local(g, [&](int i) -> int {
            const int r = g->first_rank + i;
            int rc_i;
            if ((rc_i = group_bind(g, i))) return rc_i;
            if ((rc_i = ensure_dev(g, i, dev_bytes))) return rc_i;   // (no exchange buffer: nothing to report through)
            if ((rc_i = ensure_host(g, i, pin_res_off + (i == 0 ? (list_words + flag_words) * 8 : (size_t)(nq + 1) * 8)))) return rc_i;
            char *base = reinterpret_cast<char *>(g->buf[i].dev);
            char *pin = reinterpret_cast<char *>(g->buf[i].pinned);
            uint64_t *loc = reinterpret_cast<uint64_t *>(base + loc_off);
            // a pipelined smt_sharded_search_topk_device (rccl / copy transport) may still be gathering out of this buffer on the aux stream
            if ((rc_i = drain_async(g->ctx[i]))) return rc_i;
            const int stage_rc = [&]() -> int {
                memcpy(pin, queries, (size_t)nq * SMT_DIM * 4);
                SMT_HIP_CHECK(hipMemcpyAsync(base, pin, (size_t)nq * SMT_DIM * 4, hipMemcpyHostToDevice, g->ctx[i]->stream));
                std::vector<smt_range> lr;
                if (n_ranges) layout_localize(sc, r, ranges, n_ranges, lr);
                int rc2 = search_topk_packed_local(sc->shard[i], reinterpret_cast<const float *>(base), nq, K, ws, thr_score, lr.data(),
                                                   (uint32_t)lr.size(), n_ranges != 0, sc->contiguous ? sc->rank_base[r] : 0, loc,
                                                   loc + list_words, false);
                if (!rc2 && !sc->contiguous) rc2 = layout_translate_packed(sc, i, g->ctx[i]->stream, loc, nq, K);
                if (!rc2 && i == 0) rc2 = group_debug_fail(g, SMT_DEBUG_FAIL_STAGE);
                return rc2;
            }();
            if (stage_rc) { stage_rcs[i] = stage_rc; stage_errs[i] = smt_last_error(); }
            uint64_t *status = reinterpret_cast<uint64_t *>(pin + pin_status_off);   // lives in the pinned buffer until group_sync_all
            *status = (uint64_t)(uint32_t)(stage_rc < 0 ? -stage_rc : stage_rc);
            SMT_HIP_CHECK(hipMemcpyAsync(loc + list_words + nq, status, 8, hipMemcpyHostToDevice, g->ctx[i]->stream));
            if (peer) {
                uint64_t *my_flags = reinterpret_cast<uint64_t *>(pin + pin_res_off) + (i == 0 ? list_words : 0);
                SMT_HIP_CHECK(hipMemcpyAsync(my_flags, loc + list_words, (size_t)(nq + 1) * 8, hipMemcpyDeviceToHost, g->ctx[i]->stream));
                if ((rc_i = peer_publish(g, i, g->ctx[i]->stream))) return rc_i;
            }
            return SMT_OK;
        }, g->workers != nullptr);
        if (rc) return rc;
        for (int i = 0; i < g->n_local; ++i)
            if (stage_rcs[i] && !local_rc) { local_rc = stage_rcs[i]; local_err = stage_errs[i]; }
        if (!peer && (rc = allgather_words(g, loc_off, gath_off, rank_words))) return rc;
        // the caller is one host thread and needs ONE copy of the answer: merge on local device 0
        if ((rc = group_bind(g, 0))) return rc;
        char *base0 = reinterpret_cast<char *>(g->buf[0].dev);
        uint64_t *gath = reinterpret_cast<uint64_t *>(base0 + gath_off), *merged = reinterpret_cast<uint64_t *>(base0 + out_off);
        uint64_t *h = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(g->buf[0].pinned) + pin_res_off);
        if (peer) {
            // (this entry point ends with every stream drained: the next call may overwrite the lists without asking)
            std::vector<void *> bases(g->n_local);
            for (int j = 0; j < g->n_local; ++j) bases[j] = g->buf[j].dev;
            if (!local_rc && (rc = peer_merge(g, 0, g->ctx[0]->stream, bases.data(), loc_off, nq, K, K, merged, nullptr))) return rc;
        } else {
            if (!local_rc && (rc = launch_merge_topk_packed_on(g->ctx[0], g->ctx[0]->stream, gath, (uint32_t)g->n_ranks, nq, K, K, merged, rank_words)))
                return rc;
            for (int r = 0; r < g->n_ranks; ++r)  // every rank's flags + status: all processes take the same decisions
                SMT_HIP_CHECK(hipMemcpyAsync(h + list_words + (size_t)r * (nq + 1), gath + (size_t)r * rank_words + list_words, (size_t)(nq + 1) * 8,
                                             hipMemcpyDeviceToHost, g->ctx[0]->stream));
        }
        if (!local_rc) SMT_HIP_CHECK(hipMemcpyAsync(h, merged, list_words * 8, hipMemcpyDeviceToHost, g->ctx[0]->stream));
        if ((rc = group_sync_all(g))) return rc;
        if (local_rc) { set_error("%s", local_err.c_str()); return local_rc; }
        // rank r's [nq uncertain flags][status]: gathered into device 0's block, or (peer) in rank r's own pinned block
        auto flags_of = [&](int r) -> const uint64_t * {
            if (!peer || r == 0) return h + list_words + (size_t)r * (nq + 1);
            return reinterpret_cast<const uint64_t *>(reinterpret_cast<const char *>(g->buf[r].pinned) + pin_res_off);
        };
        for (int r = 0; r < g->n_ranks; ++r) {
            const uint64_t st = flags_of(r)[nq];
            if (st) { set_error("rank %d failed with status -%llu; this rank gives up with it", r, (unsigned long long)st); return -(int)st; }
        }
        for (uint32_t q = 0; q < nq; ++q) {
            bool uncertain = false;
            for (int r = 0; r < g->n_ranks; ++r) uncertain |= flags_of(r)[q] != 0;
            if (uncertain) { redo.push_back(q); continue; }
            const uint64_t *rws = h + (size_t)q * 2 * K, *bits = rws + K;
            for (uint32_t e = 0; e < K && rws[e] != UINT64_MAX; ++e) {
                double d;
                memcpy(&d, bits + e, 8);
                hits[q].rows.push_back(rws[e]);
                hits[q].dist.push_back(d);
            }
        }
    } else {
        for (uint32_t q = 0; q < nq; ++q) redo.push_back(q);
    }
    if (!redo.empty()) {
        // threshold mode, top_k > 56, or queries whose exactness certificate failed on some shard: per-shard host
        // lists (each shard's own search is exact, fallback included), exchanged at their true sizes
        std::vector<float> sub((size_t)redo.size() * SMT_DIM);
        for (size_t j = 0; j < redo.size(); ++j) memcpy(&sub[j * SMT_DIM], queries + (size_t)redo[j] * SMT_DIM, SMT_DIM * 4);
        std::vector<std::vector<LocalHits>> local;
        rc = local_host_search(sc, sub.data(), (uint32_t)redo.size(), top_k, max_distance, mode, ranges, n_ranges, local);
        if ((rc = group_agree(g, rc))) return rc;   // (a rank whose shard search failed must not leave the others in the exchange)
        std::vector<LocalHits> merged;
        if ((rc = exchange_host_lists(g, local, (uint32_t)redo.size(), all_under_threshold ? UINT64_MAX : (uint64_t)top_k, merged)))
            return rc;
        for (size_t j = 0; j < redo.size(); ++j) hits[redo[j]] = std::move(merged[j]);
    }
    if (mode == SMT_MODE_WORKSPACE)   // (a zero query's answer is a constant: search.cpp workspace_zero_query_hits -- the same on every rank)
        for (uint32_t q = 0; q < nq; ++q)
            if (query_is_zero(queries + (size_t)q * SMT_DIM))
                workspace_zero_query_hits(ranges, n_ranges, total, top_k, has_thr, max_distance, 0, hits[q]);
    return deliver_hits(hits, out_rows, out_dist, out_counts, out_cap);
} catch (...) { return smt::api_catch(); }

int smt_sharded_search_topk_device(smt_sharded_corpus *sc, const float *const *queries_dev, uint32_t nq, uint32_t top_k,
                                   uint64_t *const *out_packed)
try {
    return smt_sharded_search_topk_device_ex(sc, queries_dev, nq, top_k, out_packed, nullptr);
} catch (...) { return smt::api_catch(); }

int smt_sharded_search_topk_device_ex(smt_sharded_corpus *sc, const float *const *queries_dev, uint32_t nq, uint32_t top_k,
                                      uint64_t *const *out_packed, uint32_t *const *out_status)
try {
    SMT_REQUIRE(sc && queries_dev && out_packed, "null argument");
    SMT_REQUIRE(top_k >= 1 && top_k <= SCAN_MAX_K, "top_k must be in [1, 56]");
    smt_group *g = sc->group;
    SMT_REQUIRE((uint64_t)g->n_ranks * top_k <= 8192, "device merge handles up to 8192 candidates per query");
    if (nq == 0) return SMT_OK;
    // with per-query verdicts wanted, every rank's nq status words (SMT_STATUS_* codes written by its select) travel behind its lists
    bool want_status = false;
    for (int i = 0; out_status && i < g->n_local; ++i) want_status = want_status || out_status[i] != nullptr;
    const size_t list_words = (size_t)nq * 2 * top_k, rank_words = list_words + (want_status ? nq : 0);
    const size_t gath_off = (rank_words * 8 + 255) & ~(size_t)255;
    const size_t dev_bytes = gath_off + (size_t)g->n_ranks * rank_words * 8 + 64;
    std::vector<char> on_aux(g->n_local, 0);
    const bool peer = g->transport == SMT_TRANSPORT_PEER;
    int rc;
    for (int i = 0; i < g->n_local; ++i) SMT_REQUIRE(queries_dev[i] != nullptr, "queries_dev");
    // the select of a single query may run on the aux stream while the NEXT call's scan streams (async select); the exchange and
    // the merge then follow it there, and the main stream carries nothing but scans
    for (int i = 0; i < g->n_local; ++i) on_aux[i] = g->ctx[i]->tune.async_select && nq == 1 && sc->shard[i]->rows >= top_k ? 1 : 0;
    const bool spread = peer && g->spread_waits && g->workers != nullptr && g->n_local <= 64;
    std::vector<std::atomic<uint64_t>> awaiting(g->n_local);   // [m]: ranks whose list is published and whose wait on m's stream nobody has enqueued yet
    std::vector<std::atomic<int>> issued(g->n_local);          // [m]: m's own scan + select + publish are on its stream
    for (int i = 0; i < g->n_local; ++i) { awaiting[i].store(0, std::memory_order_relaxed); issued[i].store(0, std::memory_order_relaxed); }
    int slot = 0;
    if (peer) {   // rank j writes its list into slot `slot` of its ring; whoever needs the answer reads the slots in place
        if ((rc = ring_ensure(g, rank_words * 8))) return rc;
        if ((rc = ring_next_slot(g, &slot))) return rc;
    }
    // every device's scan + select is issued by its own thread (group_for_each_local); the collective follows on this one
    rc = group_for_each_local(g, [&](int i) -> int {
        const int r = g->first_rank + i;
        int rc_i;
        if ((rc_i = group_bind(g, i))) return rc_i;
        if (!peer && (rc_i = ensure_dev(g, i, dev_bytes))) return rc_i;
        smt_ctx *c = g->ctx[i];
        const bool async = on_aux[i] != 0;
        uint64_t *list = peer ? reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(g->ring.dev[i]) + (size_t)slot * g->ring.slot_bytes)
                              : reinterpret_cast<uint64_t *>(g->buf[i].dev);
        rc_i = search_topk_packed_local(sc->shard[i], queries_dev[i], nq, top_k, 0, 0.f, nullptr, 0, false,
                                        sc->contiguous ? sc->rank_base[r] : 0, list, want_status ? list + list_words : nullptr, async);
        if (rc_i) return rc_i;
        if (!sc->contiguous && (rc_i = layout_translate_packed(sc, i, async ? c->aux_stream : c->stream, list, nq, top_k)))
            return rc_i;
        if (async) c->async_pending = true;
        if (out_packed[i]) prof_begin_on(c, "exchange", async ? c->aux_stream : c->stream);   // this rank's list is ready: the wait for the others starts
        if (peer && (rc_i = peer_publish(g, i, async ? c->aux_stream : c->stream))) return rc_i;
        if (spread) {
            // Every merge that will want this rank's list waits for it, enqueued off the caller's thread -- but never AHEAD of the
            // merging device's own scan: a wait for rank i's event that lands on device m's stream before m's issuer has put its scan
            // and select there makes shard m's whole share queue up behind shard i's (twice the latency, by timing).  So a
            // wait is enqueued by whoever comes SECOND: rank i posts its bit in awaiting[m] and then looks at issued[m]; if m's own
            // work is on its stream already, i takes the bit back and, if it still had it, enqueues the wait itself; m, once its own
            // work is issued, raises issued[m] and enqueues the waits of every bit it finds.  Each bit is claimed exactly once.
            const uint64_t mine = 1ull << i;
            auto stream_of = [&](int m) { return on_aux[m] ? g->ctx[m]->aux_stream : g->ctx[m]->stream; };
            if (out_packed[i]) {
                if ((rc_i = peer_await(g, i, i, stream_of(i)))) return rc_i;   // (its own list: stream order, or the hop from its aux stream)
                issued[i].store(1, std::memory_order_seq_cst);
                uint64_t got = awaiting[i].exchange(0, std::memory_order_seq_cst);
                for (int j = 0; got; ++j, got >>= 1)
                    if ((got & 1) && (rc_i = peer_await(g, j, i, stream_of(i)))) return rc_i;
            }
            for (int m = 0; m < g->n_local; ++m) {
                if (!out_packed[m] || m == i) continue;
                awaiting[m].fetch_or(mine, std::memory_order_seq_cst);
                if (issued[m].load(std::memory_order_seq_cst) && (awaiting[m].fetch_and(~mine, std::memory_order_seq_cst) & mine))
                    if ((rc_i = peer_await(g, i, m, stream_of(m)))) return rc_i;
            }
        }
        return SMT_OK;
    }, g->workers != nullptr);
    if (spread && !rc)   // (a merging device whose issuer failed before raising its flag leaves bits behind: the call fails anyway)
        for (int m = 0; m < g->n_local; ++m)
            if (uint64_t left = awaiting[m].exchange(0, std::memory_order_seq_cst))
                for (int j = 0; left; ++j, left >>= 1)
                    if ((left & 1) && (rc = peer_await(g, j, m, on_aux[m] ? g->ctx[m]->aux_stream : g->ctx[m]->stream))) return rc;
    if (rc) return rc;
    if (!peer && (rc = allgather_words(g, 0, gath_off, rank_words, &on_aux))) return rc;
    for (int i = 0; i < g->n_local; ++i) {
        uint32_t *status_i = out_status ? out_status[i] : nullptr;
        SMT_REQUIRE(!status_i || out_packed[i], "a device that wants the verdicts takes the answer too");
        if (!out_packed[i]) continue;
        smt_ctx *c = g->ctx[i];
        hipStream_t st = on_aux[i] ? c->aux_stream : c->stream;
        if (peer) {
            // (an answer wanted on ONE device -- the one-thread caller of SURVEY 8(b) -- costs n - 1 waits + a launch + a record here)
            hipEvent_t done = nullptr;
            if ((rc = ring_done_event(g, slot, i, &done))) return rc;
            const size_t off = (size_t)slot * g->ring.slot_bytes;
            // (the verdicts are read in place like the lists, by a second small kernel between the merge and the slot's `done` record)
            if ((rc = peer_merge(g, i, st, g->ring.dev.data(), off, nq, top_k, top_k, out_packed[i], status_i ? nullptr : done, spread))) return rc;
            if (status_i) {
                MergeSources src;
                for (int j = 0; j < g->n_local; ++j)
                    src.list[j] = reinterpret_cast<const uint64_t *>(reinterpret_cast<const char *>(g->ring.dev[j]) + off) + list_words;
                if ((rc = launch_combine_status_on(st, &src, nullptr, 0, (uint32_t)g->n_local, nq, status_i))) return rc;
                SMT_HIP_CHECK(hipEventRecord(done, st));
            }
            if (on_aux[i]) c->async_pending = true;
            continue;
        }
        if ((rc = group_bind(g, i))) return rc;
        const uint64_t *gath = reinterpret_cast<const uint64_t *>(reinterpret_cast<char *>(g->buf[i].dev) + gath_off);
        prof_end_on(c, "exchange", st);
        prof_begin_on(c, "merge", st);
        rc = launch_merge_topk_packed_on(c, st, gath, (uint32_t)g->n_ranks, nq, top_k, top_k, out_packed[i], want_status ? rank_words : 0);
        if (rc) return rc;
        prof_end_on(c, "merge", st);
        if (status_i && (rc = launch_combine_status_on(st, nullptr, gath + list_words, rank_words, (uint32_t)g->n_ranks, nq, status_i))) return rc;
    }
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

/* ------------------------------------------------------------ sharded IVF ---- */

int smt_sharded_ivfpq_build(smt_sharded_corpus *sc, const smt_ivfpq_params *params, int shared_centroids, smt_sharded_ivfpq **out)
try {
    SMT_REQUIRE(sc && params && out, "null argument");
    *out = nullptr;
    smt_group *g = sc->group;
    for (int r = 0; r < g->n_ranks; ++r)  // checked on EVERY rank's size: a rank bailing out alone would leave the others in the all-reduce
        SMT_REQUIRE(sc->rank_rows[r] >= params->nlist, "every shard needs at least nlist rows");
    smt_sharded_ivfpq *six = new (std::nothrow) smt_sharded_ivfpq();
    if (!six) { set_error("out of host memory"); return SMT_E_NOMEM; }
    six->corpus = sc;
    six->shared_centroids = shared_centroids ? 1 : 0;
    six->shard.assign(g->n_local, nullptr);
    std::vector<int> rcs(g->n_local, SMT_OK);
    std::vector<std::string> errs(g->n_local);
    std::vector<ShareCtx> sctx(g->n_local);
    // one host thread per local rank: the all-reduce inside the k-means loop needs every rank in it at once
    auto work = [&](int i) {
        IvfBuildShare share;
        share.rank = (uint32_t)(g->first_rank + i);
        share.n_ranks = (uint32_t)g->n_ranks;
        share.allreduce = group_allreduce_sums;
        share.agree = group_share_agree;
        sctx[i] = ShareCtx{g, i};
        share.user = &sctx[i];
        rcs[i] = ivfpq_build_shared(sc->shard[i], params, shared_centroids ? &share : nullptr, &six->shard[i]);
        if (rcs[i]) errs[i] = smt_last_error();
    };
    if (g->n_local == 1) work(0);
    else if (!run_on_threads(g->n_local, work)) { smt_sharded_ivfpq_destroy(six); return SMT_E_NOMEM; }
    for (int i = 0; i < g->n_local; ++i)
        if (rcs[i]) {
            set_error("shard %d: %s", g->first_rank + i, errs[i].c_str());
            const int rc = rcs[i];
            smt_sharded_ivfpq_destroy(six);
            return rc;
        }
    *out = six;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

void smt_sharded_ivfpq_destroy(smt_sharded_ivfpq *six)
{
    if (!six) return;
    for (smt_ivfpq *ix : six->shard) smt_ivfpq_destroy(ix);
    delete six;
}

smt_ivfpq *smt_sharded_ivfpq_shard(smt_sharded_ivfpq *six, int local_index)
{
    if (!six || local_index < 0 || local_index >= (int)six->shard.size()) return nullptr;
    return six->shard[local_index];
}

int smt_sharded_ivfpq_search(smt_sharded_ivfpq *six, const float *queries, uint32_t nq, uint32_t top_k, uint32_t nprobe, uint32_t rerank,
                             uint64_t *out_rows, double *out_dist, uint64_t *out_counts, uint64_t out_cap)
try {
    SMT_REQUIRE(six != nullptr, "index");
    SMT_REQUIRE(nq == 0 || (queries && out_rows && out_dist && out_counts), "null argument");
    SMT_REQUIRE(top_k >= 1 && top_k <= SCAN_MAX_K, "top_k must be in [1, 56]");
    smt_sharded_corpus *sc = six->corpus;
    smt_group *g = sc->group;
    SMT_REQUIRE((uint64_t)g->n_ranks * top_k <= 8192, "device merge handles up to 8192 candidates per query");
    if (nq == 0) return SMT_OK;
    if (int rcq = require_queries_domain_host(queries, nq, "smt_sharded_ivfpq_search")) return rcq;   // (domain.hip; SPMD: all ranks alike)
    if (g->n_ranks == 1) return smt_ivfpq_search(six->shard[0], queries, nq, top_k, nprobe, rerank, 0, out_rows, out_dist, out_counts, out_cap);
    const size_t list_words = (size_t)nq * 2 * top_k;
    const size_t q_bytes = ((size_t)nq * SMT_DIM * 4 + 255) & ~(size_t)255;
    const size_t loc_off = q_bytes, gath_off = loc_off + ((list_words * 8 + 255) & ~(size_t)255);
    const size_t out_off = gath_off + (((size_t)g->n_ranks * list_words * 8 + 255) & ~(size_t)255);
    const bool peer = g->transport == SMT_TRANSPORT_PEER && (uint64_t)g->n_ranks * top_k <= 4096;
    int rc;
    for (int i = 0; i < g->n_local; ++i) {
        const int r = g->first_rank + i;
        if ((rc = group_bind(g, i))) return rc;
        if ((rc = ensure_dev(g, i, out_off + list_words * 8 + 64))) return rc;
        char *base = reinterpret_cast<char *>(g->buf[i].dev);
        if ((rc = drain_async(g->ctx[i]))) return rc;   // (see smt_sharded_search: the buffer may still be read on the aux stream)
        SMT_HIP_CHECK(hipMemcpyAsync(base, queries, (size_t)nq * SMT_DIM * 4, hipMemcpyHostToDevice, g->ctx[i]->stream));
        if ((rc = ivfpq_search_packed(six->shard[i], reinterpret_cast<const float *>(base), nq, top_k, nprobe, rerank,
                                      sc->contiguous ? sc->rank_base[r] : 0, reinterpret_cast<uint64_t *>(base + loc_off))))
            return rc;
        if (!sc->contiguous && (rc = layout_translate_packed(sc, i, g->ctx[i]->stream, reinterpret_cast<uint64_t *>(base + loc_off), nq, top_k)))
            return rc;
        if (peer && (rc = peer_publish(g, i, g->ctx[i]->stream))) return rc;
    }
    if (!peer && (rc = allgather_words(g, loc_off, gath_off, list_words))) return rc;
    if ((rc = group_bind(g, 0))) return rc;
    char *base0 = reinterpret_cast<char *>(g->buf[0].dev);
    uint64_t *merged = reinterpret_cast<uint64_t *>(base0 + out_off);
    if (peer) {
        std::vector<void *> bases(g->n_local);
        for (int j = 0; j < g->n_local; ++j) bases[j] = g->buf[j].dev;
        if ((rc = peer_merge(g, 0, g->ctx[0]->stream, bases.data(), loc_off, nq, top_k, top_k, merged, nullptr))) return rc;
    } else if ((rc = launch_merge_topk_packed_on(g->ctx[0], g->ctx[0]->stream, reinterpret_cast<const uint64_t *>(base0 + gath_off),
                                                 (uint32_t)g->n_ranks, nq, top_k, top_k, merged, 0)))
        return rc;
    if ((rc = ensure_host(g, 0, list_words * 8))) return rc;
    uint64_t *h = reinterpret_cast<uint64_t *>(g->buf[0].pinned);
    SMT_HIP_CHECK(hipMemcpyAsync(h, merged, list_words * 8, hipMemcpyDeviceToHost, g->ctx[0]->stream));
    if ((rc = group_sync_all(g))) return rc;
    std::vector<LocalHits> hits(nq);
    for (uint32_t q = 0; q < nq; ++q) {
        const uint64_t *rws = h + (size_t)q * 2 * top_k, *bits = rws + top_k;
        for (uint32_t e = 0; e < top_k && rws[e] != UINT64_MAX; ++e) {
            double d;
            memcpy(&d, bits + e, 8);
            hits[q].rows.push_back(rws[e]);
            hits[q].dist.push_back(d);
        }
    }
    return deliver_hits(hits, out_rows, out_dist, out_counts, out_cap);
} catch (...) { return smt::api_catch(); }

}  // extern "C"
