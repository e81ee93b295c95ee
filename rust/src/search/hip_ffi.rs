//! FFI declarations for libsemtools_hip.so -- GENERATED from include/semtools_hip.h by tools/gen_rust_ffi.py,
//! do not edit by hand (tests/test_rust_ffi.py compares this file with the header: names, arity, scalar types).
//!
//! Uncompiled in this repository: the build container has no Rust toolchain.  A semtools maintainer adds this
//! file as `src/search/hip_ffi.rs`, the wrappers next to it (`hip.rs`, `../workspace/hip_store.rs`) and
//! `build.rs`; see INTEGRATION.md.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

macro_rules! opaque { ($($name:ident),*) => { $( #[repr(C)] pub struct $name { _private: [u8; 0] } )* } }
opaque!(SmtCtx, SmtModel, SmtCorpus, SmtIvfpq, SmtGroup, SmtShardedCorpus, SmtShardedIvfpq, SmtShardedModel);

/// half-open range of corpus rows [begin, end)
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct SmtRange {
    pub begin: u64,
    pub end: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct SmtIvfpqParams {
    pub nlist: u32,
    pub m: u32,
    pub nbits: u32,
    pub train_iters: u32,
    pub train_sample: u64,
    pub reserved: u32,
    pub local_pca: u32,
}

pub const SMT_OK: c_int = 0;
pub const SMT_E_INVALID: c_int = -1;
pub const SMT_E_HIP: c_int = -2;
pub const SMT_E_NOMEM: c_int = -3;
pub const SMT_E_TRUNCATED: c_int = -4;
pub const SMT_E_IO: c_int = -5;
pub const SMT_E_UNSUPPORTED: c_int = -6;
pub const SMT_DIM: u32 = 256;
pub const SMT_MODE_DOCUMENTS: c_int = 0;
pub const SMT_MODE_WORKSPACE: c_int = 1;
pub const SMT_STATUS_PROVED: c_int = 0;
pub const SMT_STATUS_UNCERTAIN: c_int = 1;
pub const SMT_STATUS_OVERFLOW: c_int = 2;
pub const SMT_STATUS_INVALID_QUERY: c_int = 3;
pub const SMT_UNIQUE_ID_BYTES: usize = 128;
pub const SMT_TRANSPORT_RCCL: c_int = 0;
pub const SMT_TRANSPORT_COPY: c_int = 1;
pub const SMT_TRANSPORT_PEER: c_int = 2;
pub const SMT_APPEND_WRITE_AHEAD: c_int = 1;
pub const SMT_APPEND_CREATE: c_int = 2;
pub const SMT_DEBUG_FAIL_STAGE: c_int = 1;
pub const SMT_DEBUG_FAIL_AGREE: c_int = 2;
pub const SMT_DEBUG_FAIL_BUILD: c_int = 3;

extern "C" {
    pub fn smt_ctx_create(device: c_int, out: *mut *mut SmtCtx) -> c_int;
    pub fn smt_ctx_create_on_stream(device: c_int, stream: *mut c_void, out: *mut *mut SmtCtx) -> c_int;
    pub fn smt_ctx_destroy(ctx: *mut SmtCtx);
    pub fn smt_ctx_synchronize(ctx: *mut SmtCtx) -> c_int;
    pub fn smt_last_error() -> *const c_char;
    pub fn smt_version() -> *const c_char;
    pub fn smt_device_count() -> c_int;
    pub fn smt_prof_enable(ctx: *mut SmtCtx, on: c_int) -> c_int;
    pub fn smt_prof_reset(ctx: *mut SmtCtx) -> c_int;
    pub fn smt_prof_read(ctx: *mut SmtCtx, kernel: *const c_char, launches: *mut u64, total_ms: *mut f64) -> c_int;
    pub fn smt_model_create(
        ctx: *mut SmtCtx,
        table_host: *const f32,
        V: u64,
        D: u32,
        normalize: c_int,
        out: *mut *mut SmtModel,
    ) -> c_int;
    pub fn smt_model_create_from_file(
        ctx: *mut SmtCtx,
        path: *const c_char,
        byte_offset: u64,
        V: u64,
        D: u32,
        normalize: c_int,
        out: *mut *mut SmtModel,
    ) -> c_int;
    pub fn smt_model_create_from_device(
        ctx: *mut SmtCtx,
        table_dev: *const f32,
        V: u64,
        D: u32,
        normalize: c_int,
        out: *mut *mut SmtModel,
    ) -> c_int;
    pub fn smt_model_destroy(model: *mut SmtModel);
    pub fn smt_embed(
        model: *mut SmtModel,
        ids: *const u32,
        offsets: *const u64,
        n_lines: u64,
        max_tokens: u32,
        out_host: *mut f32,
        append_to: *mut SmtCorpus,
        first_row: *mut u64,
    ) -> c_int;
    pub fn smt_embed_device(
        model: *mut SmtModel,
        ids_dev: *const u32,
        offsets_dev: *const u64,
        n_lines: u64,
        max_tokens: u32,
        out_dev: *mut f32,
    ) -> c_int;
    pub fn smt_corpus_create(ctx: *mut SmtCtx, D: u32, capacity_rows: u64, out: *mut *mut SmtCorpus) -> c_int;
    pub fn smt_corpus_from_device(
        ctx: *mut SmtCtx,
        rows_dev: *const f32,
        n_rows: u64,
        D: u32,
        out: *mut *mut SmtCorpus,
    ) -> c_int;
    pub fn smt_corpus_destroy(corpus: *mut SmtCorpus);
    pub fn smt_corpus_append_host(
        corpus: *mut SmtCorpus,
        rows: *const f32,
        n_rows: u64,
        first_row: *mut u64,
    ) -> c_int;
    pub fn smt_corpus_write_rows(corpus: *mut SmtCorpus, first_row: u64, rows: *const f32, n_rows: u64) -> c_int;
    pub fn smt_corpus_read_rows(corpus: *mut SmtCorpus, first_row: u64, n_rows: u64, out_host: *mut f32) -> c_int;
    pub fn smt_corpus_truncate(corpus: *mut SmtCorpus, n_rows: u64) -> c_int;
    pub fn smt_corpus_prepack(corpus: *mut SmtCorpus, enable: c_int) -> c_int;
    pub fn smt_corpus_image_bytes(corpus: *const SmtCorpus) -> u64;
    pub fn smt_corpus_rows(corpus: *const SmtCorpus) -> u64;
    pub fn smt_corpus_dim(corpus: *const SmtCorpus) -> u32;
    pub fn smt_corpus_save(corpus: *mut SmtCorpus, path: *const c_char) -> c_int;
    pub fn smt_corpus_load(ctx: *mut SmtCtx, path: *const c_char, out: *mut *mut SmtCorpus) -> c_int;
    pub fn smt_corpus_append_to_file(corpus: *mut SmtCorpus, path: *const c_char, rows_on_disk: u64) -> c_int;
    pub fn smt_search(
        corpus: *mut SmtCorpus,
        queries: *const f32,
        nq: u32,
        top_k: u32,
        max_distance: f64,
        mode: c_int,
        ranges: *const SmtRange,
        n_ranges: u32,
        row_base: u64,
        out_rows: *mut u64,
        out_dist: *mut f64,
        out_counts: *mut u64,
        out_cap: u64,
    ) -> c_int;
    pub fn smt_search_topk_device(
        corpus: *mut SmtCorpus,
        queries_dev: *const f32,
        nq: u32,
        top_k: u32,
        row_base: u64,
        out_rows_dev: *mut u64,
        out_dist_dev: *mut f64,
    ) -> c_int;
    pub fn smt_search_topk_device_ex(
        corpus: *mut SmtCorpus,
        queries_dev: *const f32,
        nq: u32,
        top_k: u32,
        row_base: u64,
        out_rows_dev: *mut u64,
        out_dist_dev: *mut f64,
        out_status_dev: *mut u32,
    ) -> c_int;
    pub fn smt_merge_topk(
        rows: *const u64,
        dist: *const f64,
        n_lists: u32,
        nq: u32,
        k_in: u32,
        k_out: u32,
        out_rows: *mut u64,
        out_dist: *mut f64,
        out_counts: *mut u64,
    ) -> c_int;
    pub fn smt_merge_topk_device(
        ctx: *mut SmtCtx,
        rows_dev: *const u64,
        dist_dev: *const f64,
        n_lists: u32,
        nq: u32,
        k_in: u32,
        k_out: u32,
        out_rows_dev: *mut u64,
        out_dist_dev: *mut f64,
    ) -> c_int;
    pub fn smt_merge_topk_packed_device(
        ctx: *mut SmtCtx,
        packed_dev: *const u64,
        n_lists: u32,
        nq: u32,
        k_in: u32,
        k_out: u32,
        out_packed_dev: *mut u64,
    ) -> c_int;
    pub fn smt_ivfpq_build(corpus: *mut SmtCorpus, params: *const SmtIvfpqParams, out: *mut *mut SmtIvfpq) -> c_int;
    pub fn smt_ivfpq_destroy(index: *mut SmtIvfpq);
    pub fn smt_ivfpq_search(
        index: *mut SmtIvfpq,
        queries: *const f32,
        nq: u32,
        top_k: u32,
        nprobe: u32,
        rerank: u32,
        row_base: u64,
        out_rows: *mut u64,
        out_dist: *mut f64,
        out_counts: *mut u64,
        out_cap: u64,
    ) -> c_int;
    pub fn smt_ivfpq_search_device(
        index: *mut SmtIvfpq,
        queries_dev: *const f32,
        nq: u32,
        top_k: u32,
        nprobe: u32,
        rerank: u32,
        row_base: u64,
        out_rows_dev: *mut u64,
        out_dist_dev: *mut f64,
    ) -> c_int;
    pub fn smt_ivfpq_info(
        index: *const SmtIvfpq,
        n_rows: *mut u64,
        nlist: *mut u32,
        index_bytes: *mut u64,
        build_ms4: *mut f64,
    ) -> c_int;
    pub fn smt_ivfpq_list_sizes(index: *const SmtIvfpq, sizes_host: *mut u64) -> c_int;
    pub fn smt_ivfpq_save(index: *mut SmtIvfpq, path: *const c_char) -> c_int;
    pub fn smt_ivfpq_load(corpus: *mut SmtCorpus, path: *const c_char, out: *mut *mut SmtIvfpq) -> c_int;
    pub fn smt_ivfpq_append(index: *mut SmtIvfpq, n_added: *mut u64) -> c_int;
    pub fn smt_init(devices: *const c_int, n_dev: c_int) -> c_int;
    pub fn smt_shutdown() -> c_int;
    pub fn smt_default_group() -> *mut SmtGroup;
    pub fn smt_group_create(devices: *const c_int, n_dev: c_int, out: *mut *mut SmtGroup) -> c_int;
    pub fn smt_group_create_logical(device: c_int, n_shards: c_int, out: *mut *mut SmtGroup) -> c_int;
    pub fn smt_group_from_ctx(ctx: *mut SmtCtx, out: *mut *mut SmtGroup) -> c_int;
    pub fn smt_group_unique_id(id_out: *mut c_void) -> c_int;
    pub fn smt_group_create_rank(
        device: c_int,
        rank: c_int,
        n_ranks: c_int,
        unique_id: *const c_void,
        out: *mut *mut SmtGroup,
    ) -> c_int;
    pub fn smt_group_destroy(group: *mut SmtGroup);
    pub fn smt_group_info(
        group: *const SmtGroup,
        n_ranks: *mut c_int,
        n_local: *mut c_int,
        first_rank: *mut c_int,
        rccl_ranks: *mut c_int,
        rccl_version: *mut c_int,
    ) -> c_int;
    pub fn smt_group_ctx(group: *mut SmtGroup, local_index: c_int) -> *mut SmtCtx;
    pub fn smt_group_synchronize(group: *mut SmtGroup) -> c_int;
    pub fn smt_group_barrier(group: *mut SmtGroup) -> c_int;
    pub fn smt_group_set_transport(group: *mut SmtGroup, transport: c_int) -> c_int;
    pub fn smt_group_transport(group: *const SmtGroup) -> c_int;
    pub fn smt_sharded_corpus_create(group: *mut SmtGroup, D: u32, out: *mut *mut SmtShardedCorpus) -> c_int;
    pub fn smt_sharded_corpus_from_host(
        group: *mut SmtGroup,
        rows: *const f32,
        total_rows: u64,
        D: u32,
        out: *mut *mut SmtShardedCorpus,
    ) -> c_int;
    pub fn smt_sharded_corpus_from_device(
        group: *mut SmtGroup,
        shard_rows_dev: *const *const f32,
        shard_rows: *const u64,
        D: u32,
        out: *mut *mut SmtShardedCorpus,
    ) -> c_int;
    pub fn smt_sharded_corpus_load(
        group: *mut SmtGroup,
        path: *const c_char,
        out: *mut *mut SmtShardedCorpus,
    ) -> c_int;
    pub fn smt_sharded_corpus_load_layout(
        group: *mut SmtGroup,
        path: *const c_char,
        piece_rows: *const u64,
        piece_rank: *const u32,
        n_pieces: u64,
        out: *mut *mut SmtShardedCorpus,
    ) -> c_int;
    pub fn smt_sharded_corpus_layout(
        corpus: *const SmtShardedCorpus,
        piece_rows: *mut u64,
        piece_rank: *mut u32,
        cap: u64,
    ) -> u64;
    pub fn smt_sharded_corpus_save(corpus: *mut SmtShardedCorpus, path: *const c_char) -> c_int;
    pub fn smt_sharded_corpus_append_to_file(
        corpus: *mut SmtShardedCorpus,
        path: *const c_char,
        rows_on_disk: u64,
    ) -> c_int;
    pub fn smt_sharded_corpus_append_to_file_ex(
        corpus: *mut SmtShardedCorpus,
        path: *const c_char,
        rows_on_disk: u64,
        rows_written: u64,
        flags: c_int,
    ) -> c_int;
    pub fn smt_sharded_corpus_destroy(corpus: *mut SmtShardedCorpus);
    pub fn smt_sharded_corpus_rows(corpus: *const SmtShardedCorpus) -> u64;
    pub fn smt_sharded_corpus_rank_rows(corpus: *const SmtShardedCorpus, rows_per_rank: *mut u64) -> c_int;
    pub fn smt_sharded_corpus_shard(
        corpus: *mut SmtShardedCorpus,
        local_index: c_int,
        shard: *mut *mut SmtCorpus,
        row_base: *mut u64,
        rows: *mut u64,
    ) -> c_int;
    pub fn smt_sharded_corpus_append_host(
        corpus: *mut SmtShardedCorpus,
        rows: *const f32,
        n_rows: u64,
        first_row: *mut u64,
    ) -> c_int;
    pub fn smt_sharded_corpus_read_rows(
        corpus: *mut SmtShardedCorpus,
        first_row: u64,
        n_rows: u64,
        out_host: *mut f32,
    ) -> c_int;
    pub fn smt_sharded_corpus_write_rows(
        corpus: *mut SmtShardedCorpus,
        first_row: u64,
        rows: *const f32,
        n_rows: u64,
    ) -> c_int;
    pub fn smt_sharded_model_create(
        group: *mut SmtGroup,
        table_host: *const f32,
        V: u64,
        D: u32,
        normalize: c_int,
        out: *mut *mut SmtShardedModel,
    ) -> c_int;
    pub fn smt_sharded_model_create_from_file(
        group: *mut SmtGroup,
        path: *const c_char,
        byte_offset: u64,
        V: u64,
        D: u32,
        normalize: c_int,
        out: *mut *mut SmtShardedModel,
    ) -> c_int;
    pub fn smt_sharded_model_destroy(model: *mut SmtShardedModel);
    pub fn smt_sharded_embed(
        model: *mut SmtShardedModel,
        ids: *const u32,
        offsets: *const u64,
        n_lines: u64,
        max_tokens: u32,
        out_host: *mut f32,
        append_to: *mut SmtShardedCorpus,
        first_row: *mut u64,
    ) -> c_int;
    pub fn smt_sharded_search(
        corpus: *mut SmtShardedCorpus,
        queries: *const f32,
        nq: u32,
        top_k: u32,
        max_distance: f64,
        mode: c_int,
        ranges: *const SmtRange,
        n_ranges: u32,
        out_rows: *mut u64,
        out_dist: *mut f64,
        out_counts: *mut u64,
        out_cap: u64,
    ) -> c_int;
    pub fn smt_sharded_search_topk_device(
        corpus: *mut SmtShardedCorpus,
        queries_dev: *const *const f32,
        nq: u32,
        top_k: u32,
        out_packed: *const *mut u64,
    ) -> c_int;
    pub fn smt_sharded_search_topk_device_ex(
        corpus: *mut SmtShardedCorpus,
        queries_dev: *const *const f32,
        nq: u32,
        top_k: u32,
        out_packed: *const *mut u64,
        out_status: *const *mut u32,
    ) -> c_int;
    pub fn smt_sharded_ivfpq_build(
        corpus: *mut SmtShardedCorpus,
        params: *const SmtIvfpqParams,
        shared_centroids: c_int,
        out: *mut *mut SmtShardedIvfpq,
    ) -> c_int;
    pub fn smt_sharded_ivfpq_destroy(index: *mut SmtShardedIvfpq);
    pub fn smt_sharded_ivfpq_shard(index: *mut SmtShardedIvfpq, local_index: c_int) -> *mut SmtIvfpq;
    pub fn smt_sharded_ivfpq_search(
        index: *mut SmtShardedIvfpq,
        queries: *const f32,
        nq: u32,
        top_k: u32,
        nprobe: u32,
        rerank: u32,
        out_rows: *mut u64,
        out_dist: *mut f64,
        out_counts: *mut u64,
        out_cap: u64,
    ) -> c_int;
    pub fn smt_sharded_ivfpq_save(index: *mut SmtShardedIvfpq, path: *const c_char) -> c_int;
    pub fn smt_sharded_ivfpq_load(
        corpus: *mut SmtShardedCorpus,
        path: *const c_char,
        out: *mut *mut SmtShardedIvfpq,
    ) -> c_int;
    pub fn smt_sharded_ivfpq_append(index: *mut SmtShardedIvfpq, n_added: *mut u64) -> c_int;
    pub fn smt_sharded_ivfpq_info(
        index: *const SmtShardedIvfpq,
        rows_covered: *mut u64,
        nlist: *mut u32,
        index_bytes: *mut u64,
    ) -> c_int;
    pub fn smt_ctx_uncertain_count(ctx: *mut SmtCtx, count: *mut u64, reset: c_int) -> c_int;
    pub fn smt_debug_range_sets(corpus: *const SmtCorpus, kept: *mut u64, hits: *mut u64, builds: *mut u64) -> c_int;
    pub fn smt_debug_deliveries(ctx: *mut SmtCtx, count: *mut u64) -> c_int;
    pub fn smt_debug_group_fail_next(group: *mut SmtGroup, where_: c_int, code: c_int) -> c_int;
    pub fn smt_debug_batched_scores(
        corpus: *mut SmtCorpus,
        queries: *const f32,
        nq: u32,
        first_row: u64,
        n_rows: u32,
        out: *mut f32,
    ) -> c_int;
    pub fn smt_ctx_aux_stream(ctx: *mut SmtCtx, stream_out: *mut *mut c_void) -> c_int;
    pub fn smt_set_tuning(ctx: *mut SmtCtx, key: *const c_char, value: i64) -> c_int;
    pub fn smt_fnv1a_hash(bytes: *const u8, n: u64) -> u64;
    pub fn smt_line_embedding_id(path: *const c_char, line_number: i32) -> u64;
    pub fn smt_doc_meta_id(path: *const c_char) -> u64;
}

/// `anyhow` error carrying the library's thread-local message (the reference's error type on this path).
pub fn check(rc: c_int) -> anyhow::Result<()> {
    if rc == SMT_OK {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(smt_last_error()) }.to_string_lossy().into_owned();
    anyhow::bail!("semtools_hip error {rc}: {msg}")
}
