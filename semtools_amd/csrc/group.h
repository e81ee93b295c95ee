// group.h -- internal declarations shared by group.cpp (groups, collectives, sharded search) and sharded.cpp (the
// sharded corpus' row layout and storage, the replicated model + sharded embed, the sharded index life cycle).
#pragma once
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

// One exchange workspace per local device, grown on demand.
struct GroupBuf {
    void *dev = nullptr;
    size_t dev_bytes = 0;
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
};

// Issuing threads: one per local device of a single-process group.  The caller is ONE synchronous host thread
// (src/bin/semtools.rs:134-135); issuing a search for 8 shards from it costs 8 x (bind + upload + scan + select launches) in a row --
// measured 16 us per shard (profiles/r04_group_issue.json), i.e. about one shard's whole 150 us scan at 1 M rows.  The workers issue
// every device's share at once; collectives stay on the caller's thread (ncclGroupStart / End needs all local ranks in one thread).
struct GroupWorkers {
    std::vector<std::thread> threads;
    int n_threads = 0;               // thread t issues the shares of local ranks t, t + n_threads, ...
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    const std::function<int(int)> *work = nullptr;
    // A search arrives every ~100 us in a pipelined caller: a thread that blocks in the kernel between two of them pays a futex wake
    // each way, and what that costs is the HOST's business (idle states): the same binary issued an 8-shard search in 56-67 us on
    // some leases and 85-95 on others.  So both sides SPIN first -- a worker for `spin_ns` after its last share, the caller while the
    // shares are being issued -- and only then block on the condition variables (the usual two-flag hand-over: whoever is about to
    // block says so, seq_cst, before its last look at the other side's counter).
    std::atomic<uint64_t> epoch{0};
    std::atomic<int> pending{0};
    std::atomic<int> sleepers{0};        // workers blocked (or about to block) on cv_go
    std::atomic<int> caller_blocked{0};  // the caller is blocked (or about to block) on cv_done
    std::atomic<bool> stop{false};
    long long spin_ns = 100000;          // $SEMTOOLS_GROUP_SPIN_US (0: block at once)
    std::vector<int> rcs;
    std::vector<std::string> errs;
};

struct smt_group {
    GroupWorkers *workers = nullptr; // n_local > 1: persistent issuing threads (group_for_each_local)
    int n_ranks = 0;
    int n_local = 0;
    int first_rank = 0;              // local device i is rank first_rank + i
    std::vector<smt_ctx *> ctx;      // [n_local], owned unless `borrowed`
    bool borrowed = false;           // smt_group_from_ctx: a one-rank group around a context the caller keeps
    std::vector<ncclComm_t> comm;    // [n_local]
    std::vector<GroupBuf> buf;       // [n_local]
    int rccl_version = 0;
    // Copy transport (smt_group_create_logical): every rank is a context of THIS process, possibly several on one
    // device; the all-gather is n x n device copies ordered by events instead of an RCCL collective.
    bool copies = false;
    std::vector<hipEvent_t> ev_ready;   // [n_local] rank j's send buffer is complete
    std::vector<hipEvent_t> ev_done;    // [n_local] rank i has finished reading everybody's send buffer
    // Peer transport (SMT_TRANSPORT_PEER, the default of every one-process group whose devices can read each other's memory):
    // the per-shard k-lists are not moved at all -- the merge kernel of the device that needs the answer loads them from the
    // ranks' exchange buffers where they lie (merge_topk_sources_kernel), ordered by ONE event per rank: ev_ready[j] after rank j's
    // select.  An answer costs n - 1 stream waits (enqueued by the ranks' issuers, see spread_waits) + one launch + one record; an RCCL all-gather of the same
    // 960 B costs ncclGroupStart + n ncclAllGather + ncclGroupEnd there.
    int transport = SMT_TRANSPORT_RCCL;
    bool peer_ok = false;               // every local device may read every other local device's memory
    std::vector<hipStream_t> pub_stream;   // [n_local] the stream ev_ready[j] was last recorded on
    // Who enqueues the n - 1 waits in front of a merge: the issuer of rank j, right behind its own event record (`spread_waits`: the
    // waits of one answer are spread over the issuing threads instead of queueing on the caller's thread -- ~3-5 us each), or the
    // caller's thread in peer_merge.  Checked by the self-test at creation: a runtime that refuses a wait enqueued on another
    // device's stream turns it off.  $SEMTOOLS_GROUP_WAITS=caller turns it off too (A/B).
    bool spread_waits = true;
    // The pipelined entry point (smt_sharded_search_topk_device: nothing synchronises) writes rank j's list of exchange e into slot
    // e % slots of rank j's RING, so that no rank ever waits for a reader on the device: a slot is reused `slots` exchanges later,
    // and the caller's thread first makes sure -- hipEventSynchronize on an event that has long completed, or real back-pressure
    // when the host runs `slots` searches ahead of the GPUs -- that the merges which read it are over.  (One cross-stream
    // hipStreamWaitEvent costs ~3-5 us of host time: a per-rank wait for the previous merge would double the exchange's cost.)
    struct Ring {
        std::vector<void *> dev;                    // [n_local] slots x slot_bytes
        size_t slot_bytes = 0;
        int slots = 0;
        uint64_t seq = 0;                           // exchanges issued
        std::vector<std::vector<hipEvent_t>> done;  // [slots][n_local], made on first use: device i's merge of the slot's exchange is over
        std::vector<uint64_t> merged;               // [slots] bit i: done[slot][i] was recorded for the slot's last exchange
    } ring;
    // copy-transport all-reduce (shared-centroid IVF builds run one host thread per local rank): a thread barrier
    // and the ranks' buffer addresses
    std::mutex ar_mu;
    std::condition_variable ar_cv;
    int ar_waiting = 0;
    uint64_t ar_generation = 0;
    int ar_failed = 0;               // first failure reported to group_share_agree by a rank's thread
    std::vector<const long long *> ar_sums;
    std::vector<const unsigned int *> ar_counts;
    // smt_debug_group_fail_next (test hook): the next local step of kind `debug_fail_where` on THIS process's ranks fails with
    // `debug_fail_code` -- how the tests make ONE rank of a multi-process group fail and check that all of them return together
    std::atomic<int> debug_fail_where{0};
    int debug_fail_code = 0;
};

struct smt_sharded_ivfpq {
    smt_sharded_corpus *corpus = nullptr;
    std::vector<smt_ivfpq *> shard;     // [n_local]
    int shared_centroids = 0;
};

// A run of consecutive GLOBAL rows held by one rank as consecutive LOCAL rows.
struct ShardPiece {
    uint64_t global_begin = 0, n_rows = 0, local_begin = 0;
    int rank = 0;
};

// Global rows are INSERTION ORDER over the whole corpus (the reference's tie order: document, then line --
// src/search/mod.rs:84-85,107-111).  A corpus built in one go is cut into one contiguous range per rank
// (rows_per_rank = ceil(N / n_ranks)).  A corpus that GROWS (the workspace store) deals every append over the
// ranks so that all GPUs keep embedding and scanning equal shares: the numbering is then a list of pieces.  Inside one
// rank pieces ascend in both numberings, so "local row asc" == "global row asc" there and every per-shard list is
// already in the reference's order; across ranks the merge compares (distance, global row).
struct smt_sharded_corpus {
    smt_group *group = nullptr;
    uint32_t dim = SMT_DIM;
    std::vector<smt_corpus *> shard;   // [n_local], owned handles (rows themselves may be adopted device memory)
    std::vector<uint64_t> rank_rows;   // [n_ranks]
    std::vector<uint64_t> rank_base;   // [n_ranks + 1] exclusive prefix of rank_rows; the ranks' first GLOBAL rows iff `contiguous`
    std::vector<ShardPiece> pieces;    // ascending global_begin, covering [0, total) exactly
    std::vector<std::vector<uint32_t>> rank_pieces;   // [n_ranks] indices into `pieces`, ascending (local and global)
    bool contiguous = true;            // one piece per non-empty rank, in rank order
    // device copies of the local ranks' piece tables ([n][3] u64: local_begin, n_rows, global_begin) for layout_translate_packed
    std::vector<void *> d_table;       // [n_local]
    std::vector<size_t> d_table_cap;   // [n_local] entries allocated
    std::vector<uint64_t> d_table_version;   // [n_local] layout_version the copy was made from
    uint64_t layout_version = 1;
    uint64_t total() const { return rank_base.empty() ? 0 : rank_base.back(); }
};

// The embedding table replicated on every local device ("K1 shards by line with a replicated table").
struct smt_sharded_model {
    smt_group *group = nullptr;
    std::vector<smt_model *> model;    // [n_local]
};

namespace smt {

int group_bind(smt_group *g, int i);
int ensure_dev(smt_group *g, int i, size_t bytes);
int ensure_host(smt_group *g, int i, size_t bytes);
// All-gather `words` u64 per rank: send_off / recv_off are BYTE offsets into each local device's exchange buffer.
int allgather_words(smt_group *g, size_t send_off, size_t recv_off, size_t words, const std::vector<char> *on_aux = nullptr);
int group_sync_all(smt_group *g);
int group_barrier(smt_group *g);
// Barrier that carries a status: every rank contributes `rc` and all of them return the first non-zero one (0 if none).
// Error paths of collective operations go through this so that no rank leaves while the others wait.
int group_agree(smt_group *g, int rc);
// the test hook's trigger: non-zero (an error status, message set) once if a failure of kind `where` is armed on the group
int group_debug_fail(smt_group *g, int where);
// run work(i) for every local device, on one host thread per device when there are several; first error wins
int group_for_each_local(smt_group *g, const std::function<int(int)> &work, bool threads = true);

// ---- row layout of a sharded corpus (sharded.cpp)
void layout_set_contiguous(smt_sharded_corpus *sc, const std::vector<uint64_t> &rank_rows);
// n new rows (global rows total .. total + n) dealt to the ranks: add[r] consecutive rows each, in rank order
void layout_append(smt_sharded_corpus *sc, const std::vector<uint64_t> &add);
// how an append of n rows is dealt: the emptier shards are filled first (water level), small appends go to one shard
void layout_deal(const smt_sharded_corpus *sc, uint64_t n, std::vector<uint64_t> &add);
// global ranges (sorted, disjoint) -> the LOCAL row ranges of rank r
void layout_localize(const smt_sharded_corpus *sc, int rank, const smt_range *ranges, uint32_t n, std::vector<smt_range> &out);
// local row of rank r -> global row
uint64_t layout_to_global(const smt_sharded_corpus *sc, int rank, uint64_t local_row);
// rewrite the rows of a packed [nq][2][k] list of local device i (local rows of its shard -> global rows) on `st`
int layout_translate_packed(smt_sharded_corpus *sc, int local_index, hipStream_t st, uint64_t *packed_dev, uint32_t nq, uint32_t k);

}  // namespace smt
