// corpus_io.cpp -- the corpus file (32-byte header + rows x 256 f32, rows in global order) and the
// streamed table upload: everything that moves vectors between disk and HBM through pinned double buffers.  Shared by
// smt_corpus_save / _load / _append_to_file, the sharded corpus (sharded.cpp: every rank streams its own pieces) and
// smt_model_create_from_file.
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>

#include "common.h"

using namespace smt;

struct CorpusFileHeader {  // 32 bytes, little endian
    char magic[8];         // "SMTCORP1"
    uint32_t dim;
    uint32_t reserved;
    uint64_t rows;
    uint64_t reserved2;
};


namespace smt {

// Two pinned staging buffers + one event each: the file transfer of chunk j+1 overlaps the PCIe transfer of
// chunk j (the first version read 64 K-row chunks into a pageable vector and copied them synchronously).
struct PinnedPair {
    void *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    size_t bytes = 0;
    int init(size_t want)
    {
        bytes = want;
        for (int i = 0; i < 2; ++i) {
            SMT_HIP_CHECK(hipHostMalloc(&buf[i], bytes, hipHostMallocDefault));
            SMT_HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        return SMT_OK;
    }
    int wait(int i)
    {
        if (busy[i]) { SMT_HIP_CHECK(hipEventSynchronize(ev[i])); busy[i] = false; }
        return SMT_OK;
    }
    ~PinnedPair()
    {
        for (int i = 0; i < 2; ++i) {
            if (ev[i]) { if (busy[i]) (void)hipEventSynchronize(ev[i]); (void)hipEventDestroy(ev[i]); }
            if (buf[i]) (void)hipHostFree(buf[i]);
        }
    }
};

// `bytes` from file offset `off` into `dst`, a large chunk on four threads: one thread copies out of the page cache at 10-12 GB/s,
// a fifth of what the link to the GPU takes (a warm 1 M-line workspace spent 80 ms of its 300 loading 1 GB: profiles/r04_cli_end_to_end.json)
// Returns 0, the errno of the failing pread, or -1 when the file ends early.  (The slices run on helper threads and errno is
// thread-local: each slice RETURNS its error and the first one is handed back -- and left in the caller's errno -- ADVICE r4.)
static int pread_chunk(int fd, void *dst, size_t bytes, uint64_t off)
{
    auto slice = [&](size_t b, size_t e) -> int {
        while (b < e) {
            const ssize_t got = pread(fd, static_cast<char *>(dst) + b, e - b, (off_t)(off + b));
            if (got < 0 && errno == EINTR) continue;
            if (got < 0) return errno ? errno : EIO;
            if (got == 0) return -1;
            b += (size_t)got;
        }
        return 0;
    };
    const size_t n_threads = bytes >= ((size_t)8 << 20) ? 4 : 1;
    int err[4] = {0, 0, 0, 0};
    if (n_threads == 1) err[0] = slice(0, bytes);
    else {
        std::thread th[3];
        bool started[3] = {false, false, false};
        for (size_t t = 1; t < n_threads; ++t) {
            try { th[t - 1] = std::thread([&, t] { err[t] = slice(bytes * t / n_threads, bytes * (t + 1) / n_threads); }); started[t - 1] = true; }
            catch (...) { err[t] = slice(bytes * t / n_threads, bytes * (t + 1) / n_threads); }   // (no thread to be had: read it here)
        }
        err[0] = slice(0, bytes / n_threads);
        for (size_t t = 0; t + 1 < n_threads; ++t) if (started[t]) th[t].join();
    }
    for (int e : err)
        if (e) { errno = e > 0 ? e : 0; return e; }
    return 0;
}

// the same for writing: `bytes` from `src` to file offset `off`.  (Measured on the bench box: no gain there -- persisting the 1 GB of
// rows of a 1 M-line workspace takes ~230 ms with one writer or four, i.e. the fsync at the device's ~4.4 GB/s; kept for hosts where
// the page-cache copy is the slower side.)
static int pwrite_chunk(int fd, const void *src, size_t bytes, uint64_t off)   // 0 or the errno of the failing pwrite (also left in errno)
{
    auto slice = [&](size_t b, size_t e) -> int {
        while (b < e) {
            const ssize_t put = pwrite(fd, static_cast<const char *>(src) + b, e - b, (off_t)(off + b));
            if (put < 0 && errno == EINTR) continue;
            if (put < 0) return errno ? errno : EIO;
            if (put == 0) return ENOSPC;
            b += (size_t)put;
        }
        return 0;
    };
    const size_t n_threads = bytes >= ((size_t)8 << 20) ? 4 : 1;
    int err[4] = {0, 0, 0, 0};
    if (n_threads == 1) err[0] = slice(0, bytes);
    else {
        std::thread th[3];
        bool started[3] = {false, false, false};
        for (size_t t = 1; t < n_threads; ++t) {
            try { th[t - 1] = std::thread([&, t] { err[t] = slice(bytes * t / n_threads, bytes * (t + 1) / n_threads); }); started[t - 1] = true; }
            catch (...) { err[t] = slice(bytes * t / n_threads, bytes * (t + 1) / n_threads); }
        }
        err[0] = slice(0, bytes / n_threads);
        for (size_t t = 0; t + 1 < n_threads; ++t) if (started[t]) th[t].join();
    }
    for (int e : err)
        if (e) { errno = e; return e; }
    return 0;
}

// ---------------------------------------------------------------- write-ahead: a corpus' background file writer
// SMT_APPEND_WRITE_AHEAD must not cost the caller's thread the page-cache copy (measured on the caller's thread: 41 ms per 256 MB
// batch, 166 of a 1 M-line cold workspace's 375 ms -- more than the fsync time it saved).  The caller only records "these rows
// are resident" (an event on its stream) and queues the run; ONE thread per corpus, with its own copy stream and pinned double
// buffers, copies the rows down, writes them to their place and asks the kernel to start the write-out.  It touches nothing of the
// context: the source pointer is captured when the job is queued, and whatever could move or free the rows (corpus_reserve, truncate,
// write_rows, destroy) or needs them on disk (the commit) drains the queue first (corpus_writer_drain).
struct FileWriter {
    struct Job { const float *src; uint64_t n_rows, file_row; hipEvent_t ready; std::string path; };
    std::thread th;
    std::mutex mu;
    std::condition_variable cv_work, cv_idle;
    std::deque<Job> q;
    bool stop = false, busy = false;
    int device = 0;
    uint32_t dim = 0;
    std::string error;   // first failure since the last commit (consumed by the commit's drain; the commit then writes everything itself)
    uint64_t jobs_seen = 0;

    void run()
    {
        (void)hipSetDevice(device);
        hipStream_t st = nullptr;
        PinnedPair pp;
        const size_t row_bytes = (size_t)dim * sizeof(float), chunk = 16384;   // 16 MiB copies
        bool ready = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && pp.init(chunk * row_bytes) == SMT_OK;
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(mu);
                busy = false;
                cv_idle.notify_all();
                cv_work.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) break;   // (stop: leave once the queue is empty)
                job = std::move(q.front());
                q.pop_front();
                busy = true;
            }
            std::string err;
            if (!ready) err = "write-ahead: no stream / pinned buffers";
            // test hook: $SEMTOOLS_DEBUG_FAIL_WRITE_AHEAD=n makes this writer's n-th job fail before it writes anything (what ENOSPC or
            // EIO does to a real one) -- tests/test_gpu_host.py checks that the commit still knows after the corpus has grown
            ++jobs_seen;
            if (const char *inj = getenv("SEMTOOLS_DEBUG_FAIL_WRITE_AHEAD"))
                if (err.empty() && atoll(inj) == (long long)jobs_seen) err = "write-ahead: injected failure (SEMTOOLS_DEBUG_FAIL_WRITE_AHEAD)";
            const int fd = err.empty() ? open(job.path.c_str(), O_WRONLY) : -1;
            if (err.empty() && fd < 0) err = std::string("write-ahead: cannot open '") + job.path + "': " + strerror(errno);
            if (err.empty() && hipStreamWaitEvent(st, job.ready, 0) != hipSuccess) err = "write-ahead: hipStreamWaitEvent failed";
            int j = 0;
            auto issue = [&](uint64_t r, int slot) {
                const size_t n = (size_t)std::min<uint64_t>(chunk, job.n_rows - r);
                hipError_t e = hipMemcpyAsync(pp.buf[slot], job.src + (size_t)r * dim, n * row_bytes, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipEventRecord(pp.ev[slot], st);
                if (e != hipSuccess) err = std::string("write-ahead: corpus download: ") + hipGetErrorString(e);
                else pp.busy[slot] = true;
            };
            if (err.empty()) issue(0, 0);
            for (uint64_t r = 0; r < job.n_rows && err.empty(); r += chunk, j ^= 1) {
                if (r + chunk < job.n_rows) issue(r + chunk, j ^ 1);
                if (pp.wait(j) != SMT_OK) { err = "write-ahead: hipEventSynchronize failed"; break; }
                const size_t bytes = (size_t)std::min<uint64_t>(chunk, job.n_rows - r) * row_bytes;
                const uint64_t off = sizeof(CorpusFileHeader) + (job.file_row + r) * row_bytes;
                if (const int io = pwrite_chunk(fd, pp.buf[j], bytes, off)) err = std::string("write-ahead to '") + job.path + "': " + strerror(io);
                else (void)sync_file_range(fd, (off64_t)off, (off64_t)bytes, SYNC_FILE_RANGE_WRITE);   // start the write-out, do not wait
            }
            (void)pp.wait(0);
            (void)pp.wait(1);
            if (fd >= 0) close(fd);
            (void)hipEventDestroy(job.ready);
            if (!err.empty()) {
                std::lock_guard<std::mutex> lk(mu);
                if (error.empty()) error = err;
            }
        }
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    }
};

// Wait until the writer has nothing queued or in hand.  *failed (may be null) receives its first error since the last COMMIT: only
// the caller that passes `failed` -- the commit, which then writes everything itself -- consumes it.  The other drains (the rows are
// about to move, change or go: corpus_reserve, write_rows, truncate, a save) leave it standing: a corpus doubles its capacity several
// times during an ingest, and a write-ahead job that failed before one of those growths (ENOSPC, EIO, a copy error) must still be
// known to the commit -- forgetting it would commit a header over a hole the failed job left in the file.
int corpus_writer_drain(smt_corpus *c, std::string *failed)
{
    if (failed) failed->clear();
    FileWriter *w = c->writer;
    if (!w) return SMT_OK;
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv_idle.wait(lk, [&] { return w->q.empty() && !w->busy; });
    if (failed) failed->swap(w->error);
    return SMT_OK;
}

void corpus_writer_destroy(smt_corpus *c)
{
    FileWriter *w = c->writer;
    if (!w) return;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->stop = true;
    }
    w->cv_work.notify_all();
    if (w->th.joinable()) w->th.join();
    delete w;
    c->writer = nullptr;
}

// rows [first, first + n) of `c` -> their place in `path`, in the background (the rows are resident once the context's stream
// reaches this point)
static int corpus_writer_enqueue(smt_corpus *c, const char *path, uint64_t first, uint64_t n)
{
    if (!c->writer) {
        FileWriter *w = new FileWriter();
        w->device = c->ctx->device;
        w->dim = c->dim;
        try { w->th = std::thread([w] { w->run(); }); }
        catch (...) { delete w; set_error("cannot start the write-ahead thread"); return SMT_E_NOMEM; }
        c->writer = w;
    }
    hipEvent_t ev = nullptr;
    SMT_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, c->ctx->stream);
    if (e != hipSuccess) { (void)hipEventDestroy(ev); set_error("write-ahead: %s", hipGetErrorString(e)); return SMT_E_HIP; }
    {
        std::lock_guard<std::mutex> lk(c->writer->mu);
        c->writer->q.push_back(FileWriter::Job{c->d_rows + (size_t)first * c->dim, n, first, ev, std::string(path)});
    }
    c->writer->cv_work.notify_one();
    return SMT_OK;
}

static size_t io_chunk_rows(uint64_t n_rows)
{
    // 32 MiB chunks for big files, two chunks for small ones (a 1 k-line corpus must not pin 64 MiB)
    const uint64_t big = 32768;
    return (size_t)std::max<uint64_t>(1, std::min<uint64_t>(big, (n_rows + 1) / 2));
}

int corpus_file_info(const char *path, uint64_t *rows, uint32_t *dim)
{
    FILE *f = fopen(path, "rb");
    if (!f) { set_error("cannot open '%s': %s", path, strerror(errno)); return SMT_E_IO; }
    CorpusFileHeader h;
    const bool ok = fread(&h, sizeof(h), 1, f) == 1 && memcmp(h.magic, "SMTCORP1", 8) == 0;
    bool sized = false;
    if (ok && fseek(f, 0, SEEK_END) == 0) {
        const long long sz = ftello(f);
        sized = sz >= 0 && (uint64_t)sz >= sizeof(h) + h.rows * (uint64_t)h.dim * sizeof(float);
    }
    fclose(f);
    if (!ok) { set_error("'%s' is not a corpus file", path); return SMT_E_IO; }
    if (!sized) { set_error("'%s' is truncated", path); return SMT_E_IO; }
    *rows = h.rows;
    *dim = h.dim;
    return SMT_OK;
}

// Append rows [first_row, first_row + n_rows) of the corpus file to `c`.
int corpus_load_slice(smt_corpus *c, const char *path, uint64_t first_row, uint64_t n_rows)
{
    if (n_rows == 0) return SMT_OK;
    smt_ctx *ctx = c->ctx;
    int rc = bind_device(ctx);
    if (rc) return rc;
    if ((rc = corpus_reserve(c, c->rows + n_rows))) return rc;
    FILE *f = fopen(path, "rb");
    if (!f) { set_error("cannot open '%s': %s", path, strerror(errno)); return SMT_E_IO; }
    const size_t row_bytes = (size_t)c->dim * sizeof(float);
    const uint64_t base_off = sizeof(CorpusFileHeader) + first_row * row_bytes;
    const size_t chunk = io_chunk_rows(n_rows);
    PinnedPair pp;
    if ((rc = pp.init(chunk * row_bytes))) { fclose(f); return rc; }
    int j = 0;
    for (uint64_t r = 0; r < n_rows; r += chunk, j ^= 1) {
        const size_t n = (size_t)std::min<uint64_t>(chunk, n_rows - r);
        if ((rc = pp.wait(j))) { fclose(f); return rc; }
        if (const int io = pread_chunk(fileno(f), pp.buf[j], n * row_bytes, base_off + r * row_bytes)) {
            fclose(f);
            if (io < 0) set_error("'%s' is truncated", path); else set_error("reading '%s': %s", path, strerror(io));
            return SMT_E_IO;
        }
        hipError_t e = hipMemcpyAsync(c->d_rows + (size_t)(c->rows + r) * c->dim, pp.buf[j], n * row_bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(pp.ev[j], ctx->stream);
        if (e != hipSuccess) { fclose(f); set_error("corpus upload: %s", hipGetErrorString(e)); return SMT_E_HIP; }
        pp.busy[j] = true;
    }
    fclose(f);
    SMT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    pp.busy[0] = pp.busy[1] = false;
    // a foreign or damaged file may hold anything: its rows enter the corpus only if they are in the library's domain (domain.hip)
    {
        uint64_t n_bad = 0, first = 0;
        if ((rc = check_rows_domain(ctx, c->d_rows + (size_t)c->rows * c->dim, n_rows, &n_bad, &first))) return rc;
        if (n_bad) {
            set_error("'%s': %llu rows are outside the library's domain (first: row %llu of the file) -- non-finite or absurd "
                      "magnitudes: a damaged or foreign file (include/semtools_hip.h, \"Domain\")", path, (unsigned long long)n_bad,
                      (unsigned long long)(first_row + first));
            return SMT_E_IO;   // (what a truncated file returns too: the workspace store then starts empty and re-embeds)
        }
    }
    c->rows += n_rows;
    return SMT_OK;
}

// Create `path` with a header announcing total_rows and its final size (slices are then written in place).
int corpus_file_begin(const char *path, uint32_t dim, uint64_t total_rows)
{
    FILE *f = fopen(path, "wb");
    if (!f) { set_error("cannot open '%s' for writing: %s", path, strerror(errno)); return SMT_E_IO; }
    CorpusFileHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "SMTCORP1", 8);
    h.dim = dim;
    h.rows = total_rows;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    ok = ok && fflush(f) == 0 && ftruncate(fileno(f), (off_t)(sizeof(h) + total_rows * (uint64_t)dim * sizeof(float))) == 0;
    if (fclose(f) != 0) ok = false;
    if (!ok) { set_error("cannot write '%s': %s", path, strerror(errno)); return SMT_E_IO; }
    return SMT_OK;
}

// Write runs of rows of `c` into the (existing) corpus file: run j = local rows [local_first, local_first + n_rows) at file
// row position file_first_row.  One open / fsync for all runs; the D2H of chunk j+1 flies while chunk j is written.
int corpus_save_runs(smt_corpus *c, const char *path, const FileRun *runs, size_t n_runs, bool durable)
{
    smt_ctx *ctx = c->ctx;
    int rc = bind_device(ctx);
    if (rc) return rc;
    corpus_writer_drain(c);   // (a queued write-ahead job must not write into a file that is being rewritten; its error, if any, stays for the commit)
    uint64_t total = 0, longest = 0;
    for (size_t j = 0; j < n_runs; ++j) {
        SMT_REQUIRE(runs[j].local_first + runs[j].n_rows <= c->rows, "run extends past the shard");
        total += runs[j].n_rows;
        longest = std::max(longest, runs[j].n_rows);
    }
    if (total == 0) return SMT_OK;
    const size_t chunk = io_chunk_rows(longest);
    struct Chunk { uint64_t local, n, file_row; };
    std::vector<Chunk> chunks;
    for (size_t j = 0; j < n_runs; ++j)
        for (uint64_t r = 0; r < runs[j].n_rows; r += chunk)
            chunks.push_back({runs[j].local_first + r, std::min<uint64_t>(chunk, runs[j].n_rows - r), runs[j].file_first_row + r});
    FILE *f = fopen(path, "r+b");
    if (!f) { set_error("cannot open '%s' for update: %s", path, strerror(errno)); return SMT_E_IO; }
    const size_t row_bytes = (size_t)c->dim * sizeof(float);
    PinnedPair pp;
    if ((rc = pp.init(chunk * row_bytes))) { fclose(f); return rc; }
    auto issue = [&](size_t k, int j) -> int {
        hipError_t e = hipMemcpyAsync(pp.buf[j], c->d_rows + (size_t)chunks[k].local * c->dim, (size_t)chunks[k].n * row_bytes,
                                      hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(pp.ev[j], ctx->stream);
        if (e != hipSuccess) { set_error("corpus download: %s", hipGetErrorString(e)); return SMT_E_HIP; }
        pp.busy[j] = true;
        return SMT_OK;
    };
    if ((rc = issue(0, 0))) { fclose(f); return rc; }
    int j = 0;
    int io = 0;   // the errno of the first failing step
    for (size_t k = 0; k < chunks.size() && !io; ++k, j ^= 1) {
        if (k + 1 < chunks.size() && (rc = issue(k + 1, j ^ 1))) { fclose(f); return rc; }
        if ((rc = pp.wait(j))) { fclose(f); return rc; }
        // (positional writes on the descriptor, a large chunk on four threads: nothing goes through the FILE's buffer)
        const uint64_t off = sizeof(CorpusFileHeader) + chunks[k].file_row * row_bytes;
        io = pwrite_chunk(fileno(f), pp.buf[j], (size_t)chunks[k].n * row_bytes, off);
        // write-ahead (not durable yet): ask the kernel to START writing these pages to the device now, without waiting -- the
        // commit's fsync then finds most of them on their way instead of 1 GB of dirty page cache
        if (!io && !durable) (void)sync_file_range(fileno(f), (off64_t)off, (off64_t)((size_t)chunks[k].n * row_bytes), SYNC_FILE_RANGE_WRITE);
    }
    if (!io && durable && fsync(fileno(f)) != 0) io = errno ? errno : EIO;
    if (fclose(f) != 0 && !io) io = errno ? errno : EIO;
    if (io) { set_error("short write to '%s': %s", path, strerror(io)); return SMT_E_IO; }
    return SMT_OK;
}

// Write every row of `c` into the (existing) corpus file at row position file_first_row.
int corpus_save_slice(smt_corpus *c, const char *path, uint64_t file_first_row)
{
    const FileRun run{0, c->rows, file_first_row};
    return corpus_save_runs(c, path, &run, 1);
}

// Header of an existing corpus file: check it describes `dim`-wide rows and holds exactly `expect_rows`; then (new_rows !=
// expect_rows) grow the file to new_rows WITHOUT touching the header -- corpus_file_commit writes it last, so a crash in
// between leaves the old, consistent prefix.
int corpus_file_extend(const char *path, uint32_t dim, uint64_t expect_rows, uint64_t new_rows)
{
    FILE *f = fopen(path, "r+b");
    if (!f) { set_error("cannot open '%s' for update: %s", path, strerror(errno)); return SMT_E_IO; }
    CorpusFileHeader h;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "SMTCORP1", 8) != 0 || h.dim != dim || h.rows != expect_rows) {
        fclose(f);
        set_error("'%s' does not hold exactly the first %llu rows of this corpus", path, (unsigned long long)expect_rows);
        return SMT_E_IO;
    }
    bool ok = ftruncate(fileno(f), (off_t)(sizeof(h) + new_rows * (uint64_t)dim * sizeof(float))) == 0;
    if (fclose(f) != 0) ok = false;
    if (!ok) { set_error("cannot grow '%s': %s", path, strerror(errno)); return SMT_E_IO; }
    return SMT_OK;
}

int corpus_file_commit(const char *path, uint64_t rows)
{
    FILE *f = fopen(path, "r+b");
    if (!f) { set_error("cannot open '%s' for update: %s", path, strerror(errno)); return SMT_E_IO; }
    CorpusFileHeader h;
    bool ok = fread(&h, sizeof(h), 1, f) == 1 && memcmp(h.magic, "SMTCORP1", 8) == 0;
    h.rows = rows;
    ok = ok && fseek(f, 0, SEEK_SET) == 0 && fwrite(&h, sizeof(h), 1, f) == 1 && fflush(f) == 0 && fsync(fileno(f)) == 0;
    if (fclose(f) != 0) ok = false;
    if (!ok) { set_error("cannot update the header of '%s': %s", path, strerror(errno)); return SMT_E_IO; }
    return SMT_OK;
}

// smt_[sharded_]corpus_append_to_file[_ex] on one shard: the file's header names rows_on_disk rows (checked); rows
// [rows_written, c->rows) are streamed to their places (rows_on_disk <= rows_written: what an earlier WRITE_AHEAD call already put
// there); then -- unless SMT_APPEND_WRITE_AHEAD -- fsync and the header LAST: a crash before that leaves the old, consistent prefix.
int corpus_append_to_file_ex(smt_corpus *c, const char *path, uint64_t rows_on_disk, uint64_t rows_written, int flags)
{
    SMT_REQUIRE(c != nullptr && path != nullptr, "null argument");
    SMT_REQUIRE(rows_on_disk <= rows_written && rows_written <= c->rows, "rows_on_disk <= rows_written <= rows of the corpus");
    int rc = bind_device(c->ctx);
    if (rc) return rc;
    const bool ahead = (flags & SMT_APPEND_WRITE_AHEAD) != 0;
    if (!ahead) {
        // the commit: whatever was written ahead must have reached the file; if the writer failed, everything is written here
        std::string failed;
        corpus_writer_drain(c, &failed);
        if (!failed.empty()) rows_written = rows_on_disk;
    }
    if ((flags & SMT_APPEND_CREATE) && rows_on_disk == 0 && access(path, F_OK) != 0 && (rc = corpus_file_begin(path, c->dim, 0))) return rc;
    if ((rc = corpus_file_extend(path, c->dim, rows_on_disk, c->rows))) return rc;
    if (ahead) return c->rows > rows_written ? corpus_writer_enqueue(c, path, rows_written, c->rows - rows_written) : SMT_OK;
    if (c->rows > rows_written) {
        const FileRun run{rows_written, c->rows - rows_written, rows_written};
        if ((rc = corpus_save_runs(c, path, &run, 1))) return rc;   // (its fsync covers the rows written ahead as well)
    } else if (rows_written > rows_on_disk) {
        // everything was written ahead: make it durable before the header says so
        const int fd = open(path, O_RDWR);
        const bool ok = fd >= 0 && fsync(fd) == 0;
        const int err = errno;
        if (fd >= 0) close(fd);
        if (!ok) { set_error("cannot sync '%s': %s", path, strerror(err)); return SMT_E_IO; }
    }
    return corpus_file_commit(path, c->rows);
}

}  // namespace smt

extern "C" {

int smt_corpus_save(smt_corpus *c, const char *path)
try {
    SMT_REQUIRE(c != nullptr && path != nullptr, "null argument");
    // never rewrite the live file in place: a crash or ENOSPC half way would leave a truncated corpus that
    // every later workspace command rejects.  Write a sibling, fsync, rename.
    const std::string tmp = std::string(path) + ".tmp";
    int rc = corpus_file_begin(tmp.c_str(), c->dim, c->rows);
    if (!rc) rc = corpus_save_slice(c, tmp.c_str(), 0);
    if (rc) { (void)remove(tmp.c_str()); return rc; }
    if (rename(tmp.c_str(), path) != 0) {
        set_error("rename '%s' -> '%s': %s", tmp.c_str(), path, strerror(errno));
        (void)remove(tmp.c_str());
        return SMT_E_IO;
    }
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_corpus_append_to_file(smt_corpus *c, const char *path, uint64_t rows_on_disk)
try {
    return smt::corpus_append_to_file_ex(c, path, rows_on_disk, rows_on_disk, 0);
} catch (...) { return smt::api_catch(); }

int smt_corpus_load(smt_ctx *ctx, const char *path, smt_corpus **out)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    SMT_REQUIRE(path && out, "null argument");
    *out = nullptr;
    uint64_t rows = 0;
    uint32_t dim = 0;
    if ((rc = corpus_file_info(path, &rows, &dim))) return rc;
    smt_corpus *c = nullptr;
    if ((rc = smt_corpus_create(ctx, dim, rows, &c))) return rc;
    if ((rc = corpus_load_slice(c, path, 0, rows))) { smt_corpus_destroy(c); return rc; }
    *out = c;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_model_create_from_file(smt_ctx *ctx, const char *path, uint64_t byte_offset, uint64_t V, uint32_t D, int normalize,
                               smt_model **out)
try {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    SMT_REQUIRE(out && path, "null argument");
    *out = nullptr;
    if (D != SMT_DIM) { set_error("embedding dim %u unsupported (kernels are specialised for 256)", D); return SMT_E_UNSUPPORTED; }
    SMT_REQUIRE(V > 0, "empty table");
    if ((rc = bind_device(ctx))) return rc;
    FILE *f = fopen(path, "rb");
    if (!f) { set_error("cannot open '%s': %s", path, strerror(errno)); return SMT_E_IO; }
    smt_model *m = new (std::nothrow) smt_model();
    if (!m) { fclose(f); set_error("out of host memory"); return SMT_E_NOMEM; }
    m->ctx = ctx; m->V = V; m->D = D; m->normalize = normalize ? 1 : 0; m->owned = true;
    const size_t row_bytes = (size_t)D * sizeof(float);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&m->d_table), (size_t)V * row_bytes);
    if (e != hipSuccess) { fclose(f); delete m; set_error("hipMalloc for the embedding table: %s", hipGetErrorString(e)); return SMT_E_NOMEM; }
    auto bail = [&](int code) { fclose(f); (void)hipStreamSynchronize(ctx->stream); (void)hipFree(m->d_table); delete m; return code; };
    const size_t chunk = io_chunk_rows(V);
    PinnedPair pp;
    if ((rc = pp.init(chunk * row_bytes))) return bail(rc);
    int j = 0;
    for (uint64_t r = 0; r < V; r += chunk, j ^= 1) {
        const size_t n = (size_t)std::min<uint64_t>(chunk, V - r);
        if ((rc = pp.wait(j))) return bail(rc);
        if (const int io = pread_chunk(fileno(f), pp.buf[j], n * row_bytes, byte_offset + r * row_bytes)) {
            if (io < 0) set_error("'%s' is truncated", path); else set_error("reading '%s': %s", path, strerror(io));
            return bail(SMT_E_IO);
        }
        e = hipMemcpyAsync(m->d_table + (size_t)r * D, pp.buf[j], n * row_bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(pp.ev[j], ctx->stream);
        if (e != hipSuccess) { set_error("table upload: %s", hipGetErrorString(e)); return bail(SMT_E_HIP); }
        pp.busy[j] = true;
    }
    fclose(f);
    e = hipStreamSynchronize(ctx->stream);
    pp.busy[0] = pp.busy[1] = false;
    if (e != hipSuccess) { (void)hipFree(m->d_table); delete m; set_error("table upload: %s", hipGetErrorString(e)); return SMT_E_HIP; }
    *out = m;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

}  // extern "C"
