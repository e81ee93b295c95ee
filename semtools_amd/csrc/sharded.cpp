// sharded.cpp -- what the HOST LAYER needs from a group of GPUs (VERDICT r2 row e'): a corpus that grows while staying
// balanced over the shards, persisted in the single-GPU file format; the embedding table replicated per device with a
// sharded K1 (lines dealt to the ranks in contiguous blocks, no collective); the approximate index's life cycle per shard.
//
// No reference counterpart for the sharding (the reference is one CPU process, src/bin/semtools.rs:134-135).  The callers
// these entry points serve are search_with_workspace (src/search/mod.rs:146-216), Store::upsert_line_embeddings /
// search_line_embeddings (src/workspace/store.rs:402-434, 481-546) and create_document_from_content (src/search/mod.rs:49-75),
// which see ONE matrix whose row order is insertion order; everything here keeps that view: global row == insertion order.
//
// A one-rank group is a pass-through: every entry point forwards to the single-GPU function on shard 0, so the host layer
// has ONE code path and the default single-GPU behaviour (and speed) is what it was.
#include <algorithm>
#include <cerrno>
#include <thread>

#include "group.h"

using namespace smt;

namespace smt {

// ---------------------------------------------------------------- layout
static void layout_reindex(smt_sharded_corpus *sc)
{
    const int R = sc->group->n_ranks;
    sc->rank_base.assign(R + 1, 0);
    for (int r = 0; r < R; ++r) sc->rank_base[r + 1] = sc->rank_base[r] + sc->rank_rows[r];
    sc->rank_pieces.assign(R, std::vector<uint32_t>());
    for (uint32_t k = 0; k < sc->pieces.size(); ++k) sc->rank_pieces[sc->pieces[k].rank].push_back(k);
    sc->contiguous = true;
    for (int r = 0; r < R; ++r) {
        const auto &rp = sc->rank_pieces[r];
        if (rp.empty()) continue;
        if (rp.size() > 1 || sc->pieces[rp[0]].global_begin != sc->rank_base[r]) { sc->contiguous = false; break; }
    }
    ++sc->layout_version;
}

void layout_set_contiguous(smt_sharded_corpus *sc, const std::vector<uint64_t> &rank_rows)
{
    sc->rank_rows = rank_rows;
    sc->pieces.clear();
    uint64_t gb = 0;
    for (int r = 0; r < (int)rank_rows.size(); ++r) {
        if (rank_rows[r]) {
            ShardPiece p;
            p.global_begin = gb; p.n_rows = rank_rows[r]; p.local_begin = 0; p.rank = r;
            sc->pieces.push_back(p);
        }
        gb += rank_rows[r];
    }
    layout_reindex(sc);
}

void layout_append(smt_sharded_corpus *sc, const std::vector<uint64_t> &add)
{
    uint64_t gb = sc->total();
    for (int r = 0; r < (int)add.size(); ++r) {
        if (!add[r]) continue;
        // the rank's last piece ends at its last local row; if it also ends at `gb` the new rows extend it
        bool extended = false;
        if (!sc->rank_pieces[r].empty()) {
            ShardPiece &p = sc->pieces[sc->rank_pieces[r].back()];
            if (p.global_begin + p.n_rows == gb) { p.n_rows += add[r]; extended = true; }
        }
        if (!extended) {
            ShardPiece p;
            p.global_begin = gb; p.n_rows = add[r]; p.local_begin = sc->rank_rows[r]; p.rank = r;
            sc->rank_pieces[r].push_back((uint32_t)sc->pieces.size());
            sc->pieces.push_back(p);
        }
        sc->rank_rows[r] += add[r];
        gb += add[r];
    }
    layout_reindex(sc);
}

void layout_deal(const smt_sharded_corpus *sc, uint64_t n, std::vector<uint64_t> &add)
{
    const int R = sc->group->n_ranks;
    add.assign(R, 0);
    if (n == 0) return;
    if (R == 1) { add[0] = n; return; }
    const std::vector<uint64_t> &rows = sc->rank_rows;
    if (n < (uint64_t)64 * R) {  // a handful of lines: one launch on the emptiest shard, not R slivers
        add[std::min_element(rows.begin(), rows.end()) - rows.begin()] = n;
        return;
    }
    // water level: the smallest L with sum(max(0, L - rows[r])) >= n
    uint64_t lo = *std::min_element(rows.begin(), rows.end()), hi = *std::max_element(rows.begin(), rows.end()) + n;
    auto filled = [&](uint64_t L) { uint64_t s = 0; for (uint64_t v : rows) s += L > v ? L - v : 0; return s; };
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (filled(mid) >= n) hi = mid; else lo = mid + 1;
    }
    uint64_t excess = filled(lo) - n;
    for (int r = 0; r < R; ++r) add[r] = lo > rows[r] ? lo - rows[r] : 0;
    for (int r = R - 1; r >= 0 && excess; --r)
        if (add[r]) { --add[r]; --excess; }
}

void layout_localize(const smt_sharded_corpus *sc, int rank, const smt_range *ranges, uint32_t n, std::vector<smt_range> &out)
{
    out.clear();
    uint32_t ri = 0;
    for (uint32_t idx : sc->rank_pieces[rank]) {
        const ShardPiece &p = sc->pieces[idx];
        const uint64_t pb = p.global_begin, pe = pb + p.n_rows;
        while (ri < n && ranges[ri].end <= pb) ++ri;
        for (uint32_t j = ri; j < n && ranges[j].begin < pe; ++j) {
            const uint64_t b = std::max(ranges[j].begin, pb), e = std::min(ranges[j].end, pe);
            if (e <= b) continue;
            const uint64_t lb = p.local_begin + (b - pb), le = lb + (e - b);
            if (!out.empty() && out.back().end == lb) out.back().end = le;
            else out.push_back(smt_range{lb, le});
        }
    }
}

uint64_t layout_to_global(const smt_sharded_corpus *sc, int rank, uint64_t local_row)
{
    const auto &rp = sc->rank_pieces[rank];
    size_t lo = 0, hi = rp.size();   // last piece with local_begin <= local_row
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (sc->pieces[rp[mid]].local_begin <= local_row) lo = mid; else hi = mid;
    }
    const ShardPiece &p = sc->pieces[rp[lo]];
    return p.global_begin + (local_row - p.local_begin);
}

// the piece holding global row g (g < total)
static const ShardPiece &layout_piece_of(const smt_sharded_corpus *sc, uint64_t g)
{
    size_t lo = 0, hi = sc->pieces.size();
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (sc->pieces[mid].global_begin <= g) lo = mid; else hi = mid;
    }
    return sc->pieces[lo];
}

__global__ void translate_rows_kernel(uint64_t *packed, uint32_t nq, uint32_t k, const uint64_t *tbl, uint32_t n_pieces)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * k) return;
    uint64_t *slot = packed + (size_t)(t / k) * 2 * k + (t % k);
    const uint64_t r = *slot;
    if (r == 0xFFFFFFFFFFFFFFFFull) return;
    uint32_t lo = 0, hi = n_pieces;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (tbl[3 * (size_t)mid] <= r) lo = mid; else hi = mid;
    }
    *slot = tbl[3 * (size_t)lo + 2] + (r - tbl[3 * (size_t)lo]);
}

int layout_translate_packed(smt_sharded_corpus *sc, int i, hipStream_t st, uint64_t *packed_dev, uint32_t nq, uint32_t k)
{
    smt_group *g = sc->group;
    const int r = g->first_rank + i;
    const auto &rp = sc->rank_pieces[r];
    if (rp.empty() || nq == 0 || k == 0) return SMT_OK;
    int rc = group_bind(g, i);
    if (rc) return rc;
    // (sized in sharded_new: the group's issuing threads call this for their devices at the same time)
    SMT_REQUIRE(sc->d_table.size() == (size_t)g->n_local, "sharded corpus without device tables");
    if (sc->d_table_version[i] != sc->layout_version) {
        if (sc->d_table_cap[i] < rp.size()) {
            // (the old table may still be read by a kernel in flight: free after the device has drained)
            if (sc->d_table[i]) { SMT_HIP_CHECK(hipDeviceSynchronize()); SMT_HIP_CHECK(hipFree(sc->d_table[i])); sc->d_table[i] = nullptr; }
            const size_t cap = std::max<size_t>(64, rp.size() * 2);
            SMT_HIP_CHECK(hipMalloc(&sc->d_table[i], cap * 3 * sizeof(uint64_t)));
            sc->d_table_cap[i] = cap;
        }
        std::vector<uint64_t> tbl(rp.size() * 3);
        for (size_t j = 0; j < rp.size(); ++j) {
            const ShardPiece &p = sc->pieces[rp[j]];
            tbl[3 * j] = p.local_begin; tbl[3 * j + 1] = p.n_rows; tbl[3 * j + 2] = p.global_begin;
        }
        SMT_HIP_CHECK(hipMemcpyAsync(sc->d_table[i], tbl.data(), tbl.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        SMT_HIP_CHECK(hipStreamSynchronize(st));   // `tbl` dies with this scope
        sc->d_table_version[i] = sc->layout_version;
    }
    const uint32_t n = nq * k;
    hipLaunchKernelGGL(translate_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, packed_dev, nq, k,
                       reinterpret_cast<const uint64_t *>(sc->d_table[i]), (uint32_t)rp.size());
    SMT_HIP_CHECK(hipGetLastError());
    return SMT_OK;
}

// rows_per_rank = ceil(N / n_ranks): SURVEY 8(e)
static void partition_rows(uint64_t total, int n_ranks, std::vector<uint64_t> &rank_rows)
{
    const uint64_t per = n_ranks > 0 ? (total + (uint64_t)n_ranks - 1) / (uint64_t)n_ranks : 0;
    rank_rows.assign(n_ranks, 0);
    for (int r = 0; r < n_ranks; ++r) {
        const uint64_t b = std::min<uint64_t>((uint64_t)r * per, total), e = std::min<uint64_t>((uint64_t)(r + 1) * per, total);
        rank_rows[r] = e - b;
    }
}

static smt_sharded_corpus *sharded_new(smt_group *group)
{
    smt_sharded_corpus *sc = new (std::nothrow) smt_sharded_corpus();
    if (!sc) { set_error("out of host memory"); return nullptr; }
    sc->group = group;
    sc->shard.assign(group->n_local, nullptr);
    sc->d_table.assign(group->n_local, nullptr);
    sc->d_table_cap.assign(group->n_local, 0);
    sc->d_table_version.assign(group->n_local, 0);
    layout_set_contiguous(sc, std::vector<uint64_t>(group->n_ranks, 0));
    return sc;
}

static bool single_process(const smt_group *g) { return g->n_local == g->n_ranks; }

static std::string index_shard_path(const smt_group *g, const char *path, int rank)
{
    if (g->n_ranks == 1) return path;
    return std::string(path) + ".r" + std::to_string(rank) + "of" + std::to_string(g->n_ranks);
}

}  // namespace smt

extern "C" {

/* --------------------------------------------------------- sharded corpus ---- */

void smt_sharded_corpus_destroy(smt_sharded_corpus *sc)
{
    if (!sc) return;
    for (size_t i = 0; i < sc->d_table.size(); ++i)
        if (sc->d_table[i]) { (void)group_bind(sc->group, (int)i); (void)hipDeviceSynchronize(); (void)hipFree(sc->d_table[i]); }
    for (smt_corpus *c : sc->shard) smt_corpus_destroy(c);
    delete sc;
}

int smt_sharded_corpus_create(smt_group *group, uint32_t D, smt_sharded_corpus **out)
try {
    SMT_REQUIRE(group && out, "null argument");
    *out = nullptr;
    if (D != SMT_DIM) { set_error("embedding dim %u unsupported (kernels are specialised for 256)", D); return SMT_E_UNSUPPORTED; }
    smt_sharded_corpus *sc = sharded_new(group);
    if (!sc) return SMT_E_NOMEM;
    for (int i = 0; i < group->n_local; ++i) {
        const int rc = smt_corpus_create(group->ctx[i], D, 0, &sc->shard[i]);
        if (rc) { smt_sharded_corpus_destroy(sc); return rc; }
    }
    *out = sc;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_sharded_corpus_from_host(smt_group *group, const float *rows, uint64_t total_rows, uint32_t D, smt_sharded_corpus **out)
try {
    SMT_REQUIRE(group && out, "null argument");
    *out = nullptr;
    SMT_REQUIRE(rows || total_rows == 0, "rows");
    if (D != SMT_DIM) { set_error("embedding dim %u unsupported (kernels are specialised for 256)", D); return SMT_E_UNSUPPORTED; }
    smt_sharded_corpus *sc = sharded_new(group);
    if (!sc) return SMT_E_NOMEM;
    std::vector<uint64_t> rr;
    partition_rows(total_rows, group->n_ranks, rr);
    layout_set_contiguous(sc, rr);
    for (int i = 0; i < group->n_local; ++i) {
        const int r = group->first_rank + i;
        int rc = smt_corpus_create(group->ctx[i], D, sc->rank_rows[r], &sc->shard[i]);
        if (!rc && sc->rank_rows[r])
            rc = smt_corpus_append_host(sc->shard[i], rows + (size_t)sc->rank_base[r] * D, sc->rank_rows[r], nullptr);
        if (rc) { smt_sharded_corpus_destroy(sc); return rc; }
    }
    *out = sc;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_sharded_corpus_from_device(smt_group *group, const float *const *shard_rows_dev, const uint64_t *shard_rows, uint32_t D,
                                   smt_sharded_corpus **out)
try {
    SMT_REQUIRE(group && out && shard_rows_dev && shard_rows, "null argument");
    *out = nullptr;
    if (D != SMT_DIM) { set_error("embedding dim %u unsupported", D); return SMT_E_UNSUPPORTED; }
    smt_sharded_corpus *sc = sharded_new(group);
    if (!sc) return SMT_E_NOMEM;
    int rc = SMT_OK;
    for (int i = 0; i < group->n_local && !rc; ++i)
        rc = smt_corpus_from_device(group->ctx[i], shard_rows_dev[i], shard_rows[i], D, &sc->shard[i]);
    // every rank's row count (the bases of the global row numbering): one all-gather of a word per rank
    for (int i = 0; i < group->n_local && !rc; ++i) {
        if ((rc = group_bind(group, i))) break;
        if ((rc = ensure_dev(group, i, (size_t)(1 + group->n_ranks) * 8 + 64))) break;
        hipError_t e = hipMemcpyAsync(group->buf[i].dev, &shard_rows[i], 8, hipMemcpyHostToDevice, group->ctx[i]->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(group->ctx[i]->stream);
        if (e != hipSuccess) { set_error("shard size upload: %s", hipGetErrorString(e)); rc = SMT_E_HIP; }
    }
    if (!rc) rc = allgather_words(group, 0, 8, 1);
    std::vector<uint64_t> rr(group->n_ranks, 0);
    if (!rc && !(rc = group_bind(group, 0))) {
        hipError_t e = hipMemcpyAsync(rr.data(), reinterpret_cast<char *>(group->buf[0].dev) + 8, (size_t)group->n_ranks * 8,
                                      hipMemcpyDeviceToHost, group->ctx[0]->stream);
        if (e != hipSuccess) { set_error("shard size download: %s", hipGetErrorString(e)); rc = SMT_E_HIP; }
    }
    if (!rc) rc = group_sync_all(group);
    if (rc) { smt_sharded_corpus_destroy(sc); return rc; }
    layout_set_contiguous(sc, rr);
    *out = sc;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

// Load `path` with the given piece list (global order).  Every local rank streams ITS pieces, one host thread per device.
static int sharded_load_pieces(smt_group *group, const char *path, uint32_t dim, const std::vector<std::pair<uint64_t, int>> &pieces,
                               smt_sharded_corpus **out)
{
    smt_sharded_corpus *sc = sharded_new(group);
    if (!sc) return SMT_E_NOMEM;
    std::vector<uint64_t> add(group->n_ranks, 0);
    for (auto &p : pieces) {  // one layout_append per piece: rows of one piece are consecutive on its rank
        std::fill(add.begin(), add.end(), 0);
        add[p.second] = p.first;
        layout_append(sc, add);
    }
    int rc = group_for_each_local(group, [&](int i) -> int {
        const int r = group->first_rank + i;
        int rc2 = smt_corpus_create(group->ctx[i], dim, sc->rank_rows[r], &sc->shard[i]);
        for (uint32_t idx : sc->rank_pieces[r]) {
            if (rc2) break;
            rc2 = corpus_load_slice(sc->shard[i], path, sc->pieces[idx].global_begin, sc->pieces[idx].n_rows);
        }
        return rc2;
    });
    if (rc) { smt_sharded_corpus_destroy(sc); return rc; }
    *out = sc;
    return SMT_OK;
}

int smt_sharded_corpus_load(smt_group *group, const char *path, smt_sharded_corpus **out)
try {
    SMT_REQUIRE(group && path && out, "null argument");
    *out = nullptr;
    uint64_t total = 0;
    uint32_t dim = 0;
    int rc = corpus_file_info(path, &total, &dim);
    if (rc) return rc;
    if (dim != SMT_DIM) { set_error("'%s' holds %u-dimensional rows; kernels are specialised for 256", path, dim); return SMT_E_UNSUPPORTED; }
    std::vector<uint64_t> rr;
    partition_rows(total, group->n_ranks, rr);
    std::vector<std::pair<uint64_t, int>> pieces;
    for (int r = 0; r < group->n_ranks; ++r) if (rr[r]) pieces.emplace_back(rr[r], r);
    return sharded_load_pieces(group, path, dim, pieces, out);
} catch (...) { return smt::api_catch(); }

int smt_sharded_corpus_load_layout(smt_group *group, const char *path, const uint64_t *piece_rows, const uint32_t *piece_rank,
                                   uint64_t n_pieces, smt_sharded_corpus **out)
try {
    SMT_REQUIRE(group && path && out && (n_pieces == 0 || (piece_rows && piece_rank)), "null argument");
    *out = nullptr;
    uint64_t total = 0, sum = 0;
    uint32_t dim = 0;
    int rc = corpus_file_info(path, &total, &dim);
    if (rc) return rc;
    if (dim != SMT_DIM) { set_error("'%s' holds %u-dimensional rows; kernels are specialised for 256", path, dim); return SMT_E_UNSUPPORTED; }
    std::vector<std::pair<uint64_t, int>> pieces;
    for (uint64_t k = 0; k < n_pieces; ++k) {
        SMT_REQUIRE(piece_rank[k] < (uint32_t)group->n_ranks, "piece rank outside the group");
        if (piece_rows[k]) pieces.emplace_back(piece_rows[k], (int)piece_rank[k]);
        sum += piece_rows[k];
    }
    if (sum != total) { set_error("the layout describes %llu rows, '%s' holds %llu", (unsigned long long)sum, path, (unsigned long long)total); return SMT_E_INVALID; }
    return sharded_load_pieces(group, path, dim, pieces, out);
} catch (...) { return smt::api_catch(); }

uint64_t smt_sharded_corpus_layout(const smt_sharded_corpus *sc, uint64_t *piece_rows, uint32_t *piece_rank, uint64_t cap)
{
    if (!sc) return 0;
    for (uint64_t k = 0; k < sc->pieces.size() && k < cap; ++k) {
        if (piece_rows) piece_rows[k] = sc->pieces[k].n_rows;
        if (piece_rank) piece_rank[k] = (uint32_t)sc->pieces[k].rank;
    }
    return sc->pieces.size();
}

// every local rank writes its pieces ([from_row, total) of them) into the existing file
static int sharded_write_pieces(smt_sharded_corpus *sc, const char *path, uint64_t from_row)
{
    smt_group *g = sc->group;
    return group_for_each_local(g, [&](int i) -> int {
        const int r = g->first_rank + i;
        std::vector<FileRun> runs;
        for (uint32_t idx : sc->rank_pieces[r]) {
            const ShardPiece &p = sc->pieces[idx];
            const uint64_t b = std::max(p.global_begin, from_row), e = p.global_begin + p.n_rows;
            if (e > b) runs.push_back(FileRun{p.local_begin + (b - p.global_begin), e - b, b});
        }
        return runs.empty() ? SMT_OK : corpus_save_runs(sc->shard[i], path, runs.data(), runs.size());
    });
}

int smt_sharded_corpus_save(smt_sharded_corpus *sc, const char *path)
try {
    SMT_REQUIRE(sc && path, "null argument");
    smt_group *g = sc->group;
    if (g->n_ranks == 1) return smt_corpus_save(sc->shard[0], path);
    // Write a sibling, then rename (a crash half way must not leave a truncated corpus).  Every step ends in group_agree:
    // a rank that fails (ENOSPC ...) reports it THROUGH the collective, so all ranks return the error together instead of
    // one leaving while the others wait in the next barrier.
    const std::string tmp = std::string(path) + ".tmp";
    int rc = g->first_rank == 0 ? corpus_file_begin(tmp.c_str(), sc->dim, sc->total()) : SMT_OK;
    if ((rc = group_agree(g, rc))) { if (g->first_rank == 0) (void)remove(tmp.c_str()); return rc; }
    rc = sharded_write_pieces(sc, tmp.c_str(), 0);
    if ((rc = group_agree(g, rc))) { if (g->first_rank == 0) (void)remove(tmp.c_str()); return rc; }
    if (g->first_rank == 0 && rename(tmp.c_str(), path) != 0) {
        set_error("rename '%s' -> '%s': %s", tmp.c_str(), path, strerror(errno));
        (void)remove(tmp.c_str());
        rc = SMT_E_IO;
    }
    return group_agree(g, rc);
} catch (...) { return smt::api_catch(); }

int smt_sharded_corpus_append_to_file(smt_sharded_corpus *sc, const char *path, uint64_t rows_on_disk)
try {
    SMT_REQUIRE(sc && path, "null argument");
    smt_group *g = sc->group;
    SMT_REQUIRE(rows_on_disk <= sc->total(), "file holds more rows than the corpus");
    if (g->n_ranks == 1) return smt_corpus_append_to_file(sc->shard[0], path, rows_on_disk);
    int rc = g->first_rank == 0 ? corpus_file_extend(path, sc->dim, rows_on_disk, sc->total()) : SMT_OK;
    if ((rc = group_agree(g, rc))) return rc;
    rc = sharded_write_pieces(sc, path, rows_on_disk);
    if ((rc = group_agree(g, rc))) return rc;
    if (g->first_rank == 0) rc = corpus_file_commit(path, sc->total());   // header last
    return group_agree(g, rc);
} catch (...) { return smt::api_catch(); }

int smt_sharded_corpus_append_to_file_ex(smt_sharded_corpus *sc, const char *path, uint64_t rows_on_disk, uint64_t rows_written, int flags)
try {
    SMT_REQUIRE(sc && path, "null argument");
    smt_group *g = sc->group;
    if (g->n_ranks == 1) return corpus_append_to_file_ex(sc->shard[0], path, rows_on_disk, rows_written, flags);
    // several shards: their pieces interleave in the file; writing ahead is not offered (the caller keeps everything for the commit)
    if (flags & SMT_APPEND_WRITE_AHEAD) { set_error("write-ahead appends need a one-shard corpus"); return SMT_E_UNSUPPORTED; }
    SMT_REQUIRE(rows_written == rows_on_disk, "rows written ahead on a sharded corpus");
    return smt_sharded_corpus_append_to_file(sc, path, rows_on_disk);
} catch (...) { return smt::api_catch(); }

uint64_t smt_sharded_corpus_rows(const smt_sharded_corpus *sc) { return sc ? sc->total() : 0; }

int smt_sharded_corpus_shard(smt_sharded_corpus *sc, int local_index, smt_corpus **shard, uint64_t *row_base, uint64_t *rows)
try {
    SMT_REQUIRE(sc != nullptr, "corpus");
    SMT_REQUIRE(local_index >= 0 && local_index < sc->group->n_local, "local index");
    const int r = sc->group->first_rank + local_index;
    if (shard) *shard = sc->shard[local_index];
    if (row_base) {
        // the shard's first GLOBAL row -- only a corpus cut into one range per rank has one
        SMT_REQUIRE(sc->contiguous, "this corpus grew by dealt appends: a shard holds several pieces of the global numbering (smt_sharded_corpus_layout)");
        *row_base = sc->rank_base[r];
    }
    if (rows) *rows = sc->rank_rows[r];
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_sharded_corpus_rank_rows(const smt_sharded_corpus *sc, uint64_t *rows_per_rank)
try {
    SMT_REQUIRE(sc && rows_per_rank, "null argument");
    for (int r = 0; r < sc->group->n_ranks; ++r) rows_per_rank[r] = sc->rank_rows[r];
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_sharded_corpus_append_host(smt_sharded_corpus *sc, const float *rows, uint64_t n_rows, uint64_t *first_row)
try {
    SMT_REQUIRE(sc != nullptr && (rows || n_rows == 0), "null argument");
    smt_group *g = sc->group;
    if (first_row) *first_row = sc->total();
    if (n_rows == 0) return SMT_OK;
    std::vector<uint64_t> add, begin(g->n_ranks + 1, 0);
    layout_deal(sc, n_rows, add);
    for (int r = 0; r < g->n_ranks; ++r) begin[r + 1] = begin[r] + add[r];
    int rc = SMT_OK;
    for (int i = 0; i < g->n_local && !rc; ++i) {
        const int r = g->first_rank + i;
        if (add[r]) rc = smt_corpus_append_host(sc->shard[i], rows + (size_t)begin[r] * sc->dim, add[r], nullptr);
    }
    // a rank failed -- this one, or (multi-process groups) one the agreement tells us about: every shard stays as it was
    if ((rc = group_agree(g, rc))) {
        const std::string why = smt_last_error();
        for (int i = 0; i < g->n_local; ++i) (void)smt_corpus_truncate(sc->shard[i], sc->rank_rows[g->first_rank + i]);
        set_error("%s", why.c_str());
        return rc;
    }
    layout_append(sc, add);
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

// rows [first_row, first_row + n_rows) in global order <-> host buffer.  The caller is ONE process that sees the whole
// matrix (the workspace store's compaction and in-place upserts): groups whose ranks live in other processes refuse.
static int sharded_copy_rows(smt_sharded_corpus *sc, uint64_t first_row, uint64_t n_rows, float *out_host, const float *in_host)
{
    smt_group *g = sc->group;
    if (!single_process(g)) { set_error("row access by global position needs a single-process group"); return SMT_E_UNSUPPORTED; }
    SMT_REQUIRE(first_row + n_rows <= sc->total(), "rows past the end of the corpus");
    uint64_t at = first_row;
    const uint64_t end = first_row + n_rows;
    while (at < end) {
        const ShardPiece &p = layout_piece_of(sc, at);
        const uint64_t n = std::min(end, p.global_begin + p.n_rows) - at, local = p.local_begin + (at - p.global_begin);
        const size_t off = (size_t)(at - first_row) * sc->dim;
        const int rc = out_host ? smt_corpus_read_rows(sc->shard[p.rank], local, n, out_host + off)
                                : smt_corpus_write_rows(sc->shard[p.rank], local, in_host + off, n);
        if (rc) return rc;
        at += n;
    }
    return SMT_OK;
}

int smt_sharded_corpus_read_rows(smt_sharded_corpus *sc, uint64_t first_row, uint64_t n_rows, float *out_host)
try {
    SMT_REQUIRE(sc && (out_host || n_rows == 0), "null argument");
    if (n_rows == 0) return SMT_OK;
    return sharded_copy_rows(sc, first_row, n_rows, out_host, nullptr);
} catch (...) { return smt::api_catch(); }

int smt_sharded_corpus_write_rows(smt_sharded_corpus *sc, uint64_t first_row, const float *rows, uint64_t n_rows)
try {
    SMT_REQUIRE(sc && (rows || n_rows == 0), "null argument");
    if (n_rows == 0) return SMT_OK;
    return sharded_copy_rows(sc, first_row, n_rows, nullptr, rows);
} catch (...) { return smt::api_catch(); }

/* ------------------------------------------------- replicated model + sharded K1 ---- */

void smt_sharded_model_destroy(smt_sharded_model *m)
{
    if (!m) return;
    for (smt_model *x : m->model) smt_model_destroy(x);
    delete m;
}

static int sharded_model_make(smt_group *group, const std::function<int(int, smt_model **)> &make, smt_sharded_model **out)
{
    *out = nullptr;
    smt_sharded_model *m = new (std::nothrow) smt_sharded_model();
    if (!m) { set_error("out of host memory"); return SMT_E_NOMEM; }
    m->group = group;
    m->model.assign(group->n_local, nullptr);
    // one upload per device, concurrently (each goes through that device's own pinned buffers and stream)
    int rc = group_for_each_local(group, [&](int i) -> int { return make(i, &m->model[i]); });
    if (rc) { smt_sharded_model_destroy(m); return rc; }
    *out = m;
    return SMT_OK;
}

int smt_sharded_model_create(smt_group *group, const float *table_host, uint64_t V, uint32_t D, int normalize, smt_sharded_model **out)
try {
    SMT_REQUIRE(group && table_host && out, "null argument");
    return sharded_model_make(group, [&](int i, smt_model **m) { return smt_model_create(group->ctx[i], table_host, V, D, normalize, m); }, out);
} catch (...) { return smt::api_catch(); }

int smt_sharded_model_create_from_file(smt_group *group, const char *path, uint64_t byte_offset, uint64_t V, uint32_t D, int normalize,
                                       smt_sharded_model **out)
try {
    SMT_REQUIRE(group && path && out, "null argument");
    return sharded_model_make(group, [&](int i, smt_model **m) { return smt_model_create_from_file(group->ctx[i], path, byte_offset, V, D, normalize, m); }, out);
} catch (...) { return smt::api_catch(); }

int smt_sharded_embed(smt_sharded_model *model, const uint32_t *ids, const uint64_t *offsets, uint64_t n_lines, uint32_t max_tokens,
                      float *out_host, smt_sharded_corpus *append_to, uint64_t *first_row)
try {
    SMT_REQUIRE(model != nullptr, "model");
    SMT_REQUIRE(n_lines == 0 || offsets != nullptr, "offsets");
    smt_group *g = model->group;
    if (append_to) SMT_REQUIRE(append_to->group == g, "corpus belongs to a different group");
    if (g->n_ranks == 1) {
        const int rc = smt_embed(model->model[0], ids, offsets, n_lines, max_tokens, out_host, append_to ? append_to->shard[0] : nullptr, first_row);
        if (!rc && append_to && n_lines) layout_append(append_to, std::vector<uint64_t>(1, n_lines));
        return rc;
    }
    if (first_row) *first_row = append_to ? append_to->total() : 0;
    if (n_lines == 0) return SMT_OK;
    // Lines are dealt to the ranks in contiguous blocks: block r is pooled by rank r's copy of the table and, when
    // appending, lands in rank r's shard; the blocks in rank order are the new global rows, so global row == line order.
    std::vector<uint64_t> add, begin(g->n_ranks + 1, 0);
    if (append_to) layout_deal(append_to, n_lines, add);
    else {
        // no shard to balance: equal blocks of at least 1024 lines (a query, a handful of lines: rank 0 alone)
        const uint64_t k = std::min<uint64_t>((uint64_t)g->n_ranks, std::max<uint64_t>(1, n_lines / 1024));
        add.assign(g->n_ranks, 0);
        for (uint64_t r = 0; r < k; ++r) add[r] = n_lines * (r + 1) / k - n_lines * r / k;
    }
    for (int r = 0; r < g->n_ranks; ++r) begin[r + 1] = begin[r] + add[r];
    int rc = group_for_each_local(g, [&](int i) -> int {
        const int r = g->first_rank + i;
        if (!add[r]) return SMT_OK;
        // (smt_embed rebases offsets that do not start at 0 and reads ids from ids + offsets[0])
        return smt_embed(model->model[i], ids, offsets + begin[r], add[r], max_tokens,
                         out_host ? out_host + (size_t)begin[r] * SMT_DIM : nullptr, append_to ? append_to->shard[i] : nullptr, nullptr);
    }, n_lines >= 4096);
    // a rank failed -- this one, or (multi-process groups) one the agreement tells us about: every shard stays as it was
    if ((rc = group_agree(g, rc))) {
        const std::string why = smt_last_error();
        if (append_to)
            for (int i = 0; i < g->n_local; ++i) (void)smt_corpus_truncate(append_to->shard[i], append_to->rank_rows[g->first_rank + i]);
        set_error("%s", why.c_str());
        return rc;
    }
    if (append_to) layout_append(append_to, add);
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

/* ------------------------------------------------ sharded index: life cycle ---- */

int smt_sharded_ivfpq_save(smt_sharded_ivfpq *six, const char *path)
try {
    SMT_REQUIRE(six && path, "null argument");
    smt_group *g = six->corpus->group;
    int rc = group_for_each_local(g, [&](int i) -> int {
        // a sibling first, then rename: a torn index file would be rejected by load and cost a rebuild
        const std::string file = index_shard_path(g, path, g->first_rank + i), tmp = file + ".tmp";
        int rc2 = smt_ivfpq_save(six->shard[i], tmp.c_str());
        if (!rc2 && rename(tmp.c_str(), file.c_str()) != 0) { set_error("rename '%s' -> '%s': %s", tmp.c_str(), file.c_str(), strerror(errno)); rc2 = SMT_E_IO; }
        if (rc2) (void)remove(tmp.c_str());
        return rc2;
    });
    return group_agree(g, rc);
} catch (...) { return smt::api_catch(); }

int smt_sharded_ivfpq_load(smt_sharded_corpus *sc, const char *path, smt_sharded_ivfpq **out)
try {
    SMT_REQUIRE(sc && path && out, "null argument");
    *out = nullptr;
    smt_group *g = sc->group;
    smt_sharded_ivfpq *six = new (std::nothrow) smt_sharded_ivfpq();
    if (!six) { set_error("out of host memory"); return SMT_E_NOMEM; }
    six->corpus = sc;
    six->shard.assign(g->n_local, nullptr);
    int rc = group_for_each_local(g, [&](int i) -> int {
        return smt_ivfpq_load(sc->shard[i], index_shard_path(g, path, g->first_rank + i).c_str(), &six->shard[i]);
    });
    if ((rc = group_agree(g, rc))) { smt_sharded_ivfpq_destroy(six); return rc; }
    *out = six;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_sharded_ivfpq_append(smt_sharded_ivfpq *six, uint64_t *n_added)
try {
    SMT_REQUIRE(six != nullptr, "index");
    smt_group *g = six->corpus->group;
    std::vector<uint64_t> added(g->n_local, 0);
    int rc = group_for_each_local(g, [&](int i) -> int { return smt_ivfpq_append(six->shard[i], &added[i]); });
    if ((rc = group_agree(g, rc))) return rc;
    if (n_added) { *n_added = 0; for (uint64_t a : added) *n_added += a; }   // (local ranks)
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

int smt_sharded_ivfpq_info(const smt_sharded_ivfpq *six, uint64_t *rows_covered, uint32_t *nlist, uint64_t *index_bytes)
try {
    SMT_REQUIRE(six != nullptr, "index");
    uint64_t rows = 0, bytes = 0;
    uint32_t lists = 0;
    for (smt_ivfpq *ix : six->shard) {   // (local ranks)
        uint64_t n = 0, b = 0;
        uint32_t l = 0;
        const int rc = smt_ivfpq_info(ix, &n, &l, &b, nullptr);
        if (rc) return rc;
        rows += n; bytes += b; lists = std::max(lists, l);
    }
    if (rows_covered) *rows_covered = rows;
    if (nlist) *nlist = lists;
    if (index_bytes) *index_bytes = bytes;
    return SMT_OK;
} catch (...) { return smt::api_catch(); }

}  // extern "C"
