// fake_rccl.cpp -- TEST INFRASTRUCTURE: a test double for librccl.so.1 that lets N processes form one communicator ON ONE GPU.
//
// Why: RCCL refuses two ranks on one device, and the GPU boxes the tests run on have one.  The product's one-rank-per-process
// group (semtools_amd/csrc/group.cpp: smt_group_create_rank -> ncclCommInitRank, allgather_words, group_agree, group_barrier,
// exchange_host_lists, the shared-centroid ncclAllReduce) would otherwise never execute with n_ranks > 1 before a real 8-GPU
// node runs it.  The library binds RCCL with dlopen("librccl.so.1", RTLD_NOLOAD) first (group.cpp load_rccl), so an object with
// that SONAME already mapped into the process -- LD_PRELOAD, or ctypes.CDLL(..., RTLD_GLOBAL) before the first group -- takes
// RCCL's place with NO product change.  Nothing here is linked, loaded or shipped by the product.
//
// What it keeps of RCCL's contract (the part group.cpp relies on):
//   * the twelve entry points group.cpp binds, same signatures (rccl/rccl.h);
//   * a collective is ENQUEUED on the caller's stream and returns at once: results exist only after the stream has been
//     synchronised (a double that completed inside the call would hide a missing synchronisation in the product);
//   * collectives of one communicator execute in issue order even when issued on different streams (RCCL chains them with
//     events; so does this);
//   * every rank must issue the same collectives in the same order; a rank that does not arrive strands the others -- here
//     for $FAKE_RCCL_TIMEOUT_S (default 60) seconds, after which every later call fails with ncclSystemError instead of hanging.
// What it is not: fast, or a model of xGMI.  Payloads travel device -> pinned staging -> POSIX shared memory -> pinned staging
// -> device in 1 MiB chunks, the ranks meeting in a host callback (hipLaunchHostFunc) on per-rank sequence numbers.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace {

constexpr int MAX_RANKS = 16;
constexpr size_t SLOT_BYTES = (size_t)1 << 20;   // per rank, per chunk
constexpr uint64_t MAGIC = 0x46414B455243434Cull; // "FAKERCCL"
constexpr size_t HDR_BYTES = 4096;

struct ShmHeader {
    uint64_t magic;
    std::atomic<int32_t> n_ranks;              // set by the first rank to attach, checked by the others
    std::atomic<int32_t> attached;
    std::atomic<int32_t> aborted;
    std::atomic<uint64_t> arrive[MAX_RANKS];   // arrive[r] = s: rank r's contribution to chunk s lies in slot s & 1
};
static_assert(sizeof(ShmHeader) <= HDR_BYTES, "header");
constexpr size_t SHM_BYTES = HDR_BYTES + 2 * MAX_RANKS * SLOT_BYTES;

struct UniqueIdBody {   // what travels in the 128 bytes
    char tag[8];        // "FRCCL01"
    char name[96];
};
static_assert(sizeof(UniqueIdBody) <= sizeof(ncclUniqueId), "id");

thread_local char g_err[256] = "";
std::atomic<uint64_t> g_allgathers{0}, g_allreduces{0}, g_bytes{0}, g_chunks{0};

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double timeout_s()
{
    const char *e = getenv("FAKE_RCCL_TIMEOUT_S");
    const double v = e ? atof(e) : 60.0;
    return v > 0 ? v : 60.0;
}

}  // namespace

struct ncclComm {   // (rccl.h forward-declares it: ncclComm_t = ncclComm *)
    int rank = 0, n = 0, device = 0;
    ShmHeader *hdr = nullptr;
    char *data = nullptr;                 // [2][n][SLOT_BYTES]
    char *stage_out = nullptr;            // pinned, SLOT_BYTES
    char *stage_in = nullptr;             // pinned, n * SLOT_BYTES
    uint64_t seq = 0;                     // chunks issued
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    hipEvent_t chain = nullptr;
    std::atomic<int> failed{0};
    char *slot(uint64_t s, int r) const { return data + (((s & 1) * (uint64_t)n + (uint64_t)r) * SLOT_BYTES); }
};

namespace {

enum Op { OP_GATHER = 0, OP_SUM_I64, OP_SUM_U32, OP_SUM_F32, OP_SUM_F64, OP_SUM_I32, OP_SUM_U64 };

struct Job {
    ncclComm *c;
    uint64_t s;
    size_t bytes;   // this chunk, per rank
    int op;
};

bool wait_all(ncclComm *c, uint64_t s)
{
    const double t0 = now_s(), lim = timeout_s();
    unsigned spins = 0;
    for (;;) {
        bool all = true;
        for (int r = 0; r < c->n; ++r)
            if (c->hdr->arrive[r].load(std::memory_order_acquire) < s) { all = false; break; }
        if (all) return true;
        if (c->hdr->aborted.load(std::memory_order_acquire)) return false;
        if ((++spins & 63) == 0) {
            if (now_s() - t0 > lim) {
                c->hdr->aborted.store(1, std::memory_order_release);
                fprintf(stderr, "fake_rccl: rank %d of %d waited %.0f s at chunk %llu for a rank that never arrived\n", c->rank, c->n, lim,
                        (unsigned long long)s);
                return false;
            }
            usleep(50);
        }
    }
}

template <typename T>
void sum_into(char *dst, const ncclComm *c, uint64_t s, size_t bytes)
{
    T *out = reinterpret_cast<T *>(dst);
    const size_t cnt = bytes / sizeof(T);
    for (size_t i = 0; i < cnt; ++i) out[i] = 0;
    for (int r = 0; r < c->n; ++r) {   // rank order: every rank computes the same sum, bit for bit
        const T *in = reinterpret_cast<const T *>(c->slot(s, r));
        for (size_t i = 0; i < cnt; ++i) out[i] += in[i];
    }
}

// Runs on a HIP runtime thread, in stream order behind the chunk's device -> stage_out copy.  No HIP calls in here.
void meet(void *arg)
{
    Job *j = static_cast<Job *>(arg);
    ncclComm *c = j->c;
    if (!c->failed.load()) {
        memcpy(c->slot(j->s, c->rank), c->stage_out, j->bytes);
        c->hdr->arrive[c->rank].store(j->s, std::memory_order_release);
        if (!wait_all(c, j->s)) c->failed.store(1);
        else if (j->op == OP_GATHER)
            for (int r = 0; r < c->n; ++r) memcpy(c->stage_in + (size_t)r * SLOT_BYTES, c->slot(j->s, r), j->bytes);
        else if (j->op == OP_SUM_I64) sum_into<long long>(c->stage_in, c, j->s, j->bytes);
        else if (j->op == OP_SUM_U64) sum_into<unsigned long long>(c->stage_in, c, j->s, j->bytes);
        else if (j->op == OP_SUM_U32) sum_into<unsigned int>(c->stage_in, c, j->s, j->bytes);
        else if (j->op == OP_SUM_I32) sum_into<int>(c->stage_in, c, j->s, j->bytes);
        else if (j->op == OP_SUM_F32) sum_into<float>(c->stage_in, c, j->s, j->bytes);
        else if (j->op == OP_SUM_F64) sum_into<double>(c->stage_in, c, j->s, j->bytes);
    }
    g_chunks.fetch_add(1);
    delete j;
}

ncclResult_t fail(ncclResult_t r, const char *what, hipError_t e = hipSuccess)
{
    if (e != hipSuccess) snprintf(g_err, sizeof g_err, "fake_rccl: %s: %s", what, hipGetErrorString(e));
    else snprintf(g_err, sizeof g_err, "fake_rccl: %s", what);
    return r;
}

#define FR_HIP(expr)                                                   \
    do {                                                               \
        hipError_t _e = (expr);                                        \
        if (_e != hipSuccess) return fail(ncclUnhandledCudaError, #expr, _e); \
    } while (0)

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

// Collectives of one communicator run in issue order whatever streams they were given.
ncclResult_t chain_to(ncclComm *c, hipStream_t st)
{
    if (c->have_last && c->last_stream != st) {
        FR_HIP(hipEventRecord(c->chain, c->last_stream));
        FR_HIP(hipStreamWaitEvent(st, c->chain, 0));
    }
    c->last_stream = st;
    c->have_last = true;
    return ncclSuccess;
}

ncclResult_t collective(ncclComm *c, const char *send, char *recv, size_t bytes_per_rank, int op, hipStream_t st)
{
    if (!c) return fail(ncclInvalidArgument, "null communicator");
    if (c->failed.load() || c->hdr->aborted.load()) return fail(ncclSystemError, "the communicator was aborted (a rank never arrived)");
    FR_HIP(hipSetDevice(c->device));
    ncclResult_t r = chain_to(c, st);
    if (r != ncclSuccess) return r;
    for (size_t off = 0; off < bytes_per_rank || (bytes_per_rank == 0 && off == 0); off += SLOT_BYTES) {
        const size_t nb = bytes_per_rank - off < SLOT_BYTES ? bytes_per_rank - off : SLOT_BYTES;
        const uint64_t s = ++c->seq;
        if (nb) FR_HIP(hipMemcpyAsync(c->stage_out, send + off, nb, hipMemcpyDeviceToHost, st));
        Job *j = new (std::nothrow) Job{c, s, nb, op};
        if (!j) return fail(ncclSystemError, "out of host memory");
        FR_HIP(hipLaunchHostFunc(st, meet, j));
        if (op == OP_GATHER) {
            for (int q = 0; q < c->n && nb; ++q)
                FR_HIP(hipMemcpyAsync(recv + (size_t)q * bytes_per_rank + off, c->stage_in + (size_t)q * SLOT_BYTES, nb, hipMemcpyHostToDevice, st));
        } else if (nb) {
            FR_HIP(hipMemcpyAsync(recv + off, c->stage_in, nb, hipMemcpyHostToDevice, st));
        }
        if (bytes_per_rank == 0) break;
    }
    g_bytes.fetch_add(bytes_per_rank);
    return ncclSuccess;
}

ncclResult_t attach(ncclComm **out, const char *name, int n, int rank, bool unlink_when_full)
{
    if (n < 1 || n > MAX_RANKS || rank < 0 || rank >= n) return fail(ncclInvalidArgument, "rank / nranks (the double handles up to 16 ranks)");
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return fail(ncclSystemError, "shm_open of the communicator's segment failed (was the unique id made on this host?)");
    void *p = mmap(nullptr, SHM_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail(ncclSystemError, "mmap");
    ShmHeader *h = static_cast<ShmHeader *>(p);
    if (h->magic != MAGIC) { munmap(p, SHM_BYTES); return fail(ncclInvalidArgument, "not a fake_rccl segment"); }
    int32_t expect = 0;
    if (!h->n_ranks.compare_exchange_strong(expect, n) && expect != n) { munmap(p, SHM_BYTES); return fail(ncclInvalidArgument, "ranks disagree on nranks"); }
    ncclComm *c = new (std::nothrow) ncclComm();
    if (!c) { munmap(p, SHM_BYTES); return fail(ncclSystemError, "out of host memory"); }
    c->rank = rank;
    c->n = n;
    c->hdr = h;
    c->data = static_cast<char *>(p) + HDR_BYTES;
    hipError_t e = hipGetDevice(&c->device);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->stage_out), SLOT_BYTES, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->stage_in), (size_t)n * SLOT_BYTES, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->chain, hipEventDisableTiming);
    if (e != hipSuccess) { ncclCommDestroy(c); return fail(ncclUnhandledCudaError, "staging buffers", e); }
    const int32_t now = h->attached.fetch_add(1) + 1;
    if (now == n && unlink_when_full) shm_unlink(name);   // everybody has it mapped: the name can go
    // ncclCommInitRank returns when every rank has joined
    const double t0 = now_s(), lim = timeout_s();
    while (h->attached.load(std::memory_order_acquire) < n) {
        if (now_s() - t0 > lim) {
            h->aborted.store(1);
            ncclCommDestroy(c);
            return fail(ncclSystemError, "ncclCommInitRank: the other ranks never joined");
        }
        usleep(200);
    }
    *out = c;
    return ncclSuccess;
}

ncclResult_t make_segment(char *name_out, size_t cap)
{
    const char *tag = getenv("FAKE_RCCL_TAG");
    static std::atomic<unsigned> counter{0};
    snprintf(name_out, cap, "/fake_rccl_%s_%d_%u_%llx", tag ? tag : "x", (int)getpid(), counter.fetch_add(1),
             (unsigned long long)(now_s() * 1e6));
    const int fd = shm_open(name_out, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return fail(ncclSystemError, "shm_open(O_CREAT)");
    if (ftruncate(fd, (off_t)SHM_BYTES) != 0) { close(fd); shm_unlink(name_out); return fail(ncclSystemError, "ftruncate"); }
    void *p = mmap(nullptr, HDR_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { shm_unlink(name_out); return fail(ncclSystemError, "mmap"); }
    ShmHeader *h = new (p) ShmHeader();   // (tmpfs pages start zeroed; the atomics are lock-free words)
    h->magic = MAGIC;
    munmap(p, HDR_BYTES);
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version)
{
    if (!version) return fail(ncclInvalidArgument, "version");
    *version = 22000 + 99;   // "2.20.99": recognisably not a real release
    return ncclSuccess;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *uniqueId)
{
    if (!uniqueId) return fail(ncclInvalidArgument, "uniqueId");
    memset(uniqueId, 0, sizeof *uniqueId);
    UniqueIdBody *b = reinterpret_cast<UniqueIdBody *>(uniqueId);
    memcpy(b->tag, "FRCCL01", 8);
    return make_segment(b->name, sizeof b->name);
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId commId, int rank)
{
    if (!comm) return fail(ncclInvalidArgument, "comm");
    *comm = nullptr;
    const UniqueIdBody *b = reinterpret_cast<const UniqueIdBody *>(&commId);
    if (memcmp(b->tag, "FRCCL01", 8) != 0 || memchr(b->name, 0, sizeof b->name) == nullptr)
        return fail(ncclInvalidArgument, "this unique id was not made by fake_rccl's ncclGetUniqueId");
    return attach(comm, b->name, nranks, rank, true);
}

ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist)
{
    // One process, several devices: the host callbacks of the local ranks would have to run concurrently, which the runtime does
    // not promise.  The tests use the real RCCL for one-process groups (one rank) and logical groups for n > 1.
    if (!comm || ndev != 1) return fail(ncclInvalidUsage, "the double implements ncclCommInitAll for one device only");
    char name[96];
    ncclResult_t r = make_segment(name, sizeof name);
    if (r != ncclSuccess) return r;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (devlist) FR_HIP(hipSetDevice(devlist[0]));
    r = attach(comm, name, 1, 0, true);
    (void)hipSetDevice(prev);
    return r;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    if (c->have_last) (void)hipStreamSynchronize(c->last_stream);   // no callback of ours is left behind
    if (c->chain) (void)hipEventDestroy(c->chain);
    if (c->stage_out) (void)hipHostFree(c->stage_out);
    if (c->stage_in) (void)hipHostFree(c->stage_in);
    if (c->hdr) munmap(c->hdr, SHM_BYTES);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    if (!comm || !count) return fail(ncclInvalidArgument, "null argument");
    *count = comm->n;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank)
{
    if (!comm || !rank) return fail(ncclInvalidArgument, "null argument");
    *rank = comm->rank;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    const size_t tb = type_bytes(datatype);
    if (!tb) return fail(ncclInvalidArgument, "datatype");
    g_allgathers.fetch_add(1);
    return collective(comm, static_cast<const char *>(sendbuff), static_cast<char *>(recvbuff), sendcount * tb, OP_GATHER, stream);
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
    if (op != ncclSum) return fail(ncclInvalidArgument, "the double sums only");
    int o;
    switch (datatype) {
    case ncclInt64: o = OP_SUM_I64; break;
    case ncclUint64: o = OP_SUM_U64; break;
    case ncclUint32: o = OP_SUM_U32; break;
    case ncclInt32: o = OP_SUM_I32; break;
    case ncclFloat32: o = OP_SUM_F32; break;
    case ncclFloat64: o = OP_SUM_F64; break;
    default: return fail(ncclInvalidArgument, "datatype");
    }
    g_allreduces.fetch_add(1);
    return collective(comm, static_cast<const char *>(sendbuff), static_cast<char *>(recvbuff), count * type_bytes(datatype), o, stream);
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }   // (one communicator per process: nothing to fuse)
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

const char *ncclGetErrorString(ncclResult_t result)
{
    if (result == ncclSuccess) return "no error";
    return g_err[0] ? g_err : "fake_rccl: error";
}

// For the tests: proof that the exchange really went through the double.
void fake_rccl_stats(uint64_t *allgathers, uint64_t *allreduces, uint64_t *payload_bytes, uint64_t *chunks_met)
{
    if (allgathers) *allgathers = g_allgathers.load();
    if (allreduces) *allreduces = g_allreduces.load();
    if (payload_bytes) *payload_bytes = g_bytes.load();
    if (chunks_met) *chunks_met = g_chunks.load();
}

}  // extern "C"
