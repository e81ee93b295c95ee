//! GPU-backed replacements for the bodies of the search functions of `src/search/mod.rs` (reference v3.0.0).
//! UNCOMPILED here (no Rust toolchain in the build container) -- see rust/README.md.
//!
//! What moves to the GPU: the pool step of `StaticModel::encode_with_args` / `encode_single`
//! (mod.rs:69,138,153), the `f32::cosine` loop (mod.rs:84-86) and the threshold / stable sort / take
//! (mod.rs:88-119).  What stays in Rust: tokenisation (`tokenizers`), file I/O, context windows, printing.
//!
//! A `Document` no longer owns `Vec<Vec<f32>>`: its line embeddings are rows `first_row .. first_row + lines.len()`
//! of the corpus resident in HBM, in (document, line) order -- which is exactly the order the reference's stable
//! sort breaks distance ties in, so "row ascending" reproduces it.
use super::hip_ffi::*;
use super::{SearchConfig, SearchResult};
use anyhow::Result;
use std::cmp::min;
use std::fs::read_to_string;
use std::ptr;
use tokenizers::Tokenizer;

/// Handles that live as long as the process: the GPUs this process owns (an `smt_group`: one GPU by default, every
/// GPU of the node when `devices` lists them -- the reference runs the whole search from ONE task,
/// src/bin/semtools.rs:134-135, and that one caller drives them all), the model table replicated on each of them,
/// and the corpus of this invocation, its rows dealt over the GPUs as they are embedded.  With one device every
/// `smt_sharded_*` call below IS its single-GPU counterpart (`smt_group_from_ctx`).
pub struct GpuSearch {
    ctx: *mut SmtCtx,                 // only for the one-GPU form (the group borrows it)
    group: *mut SmtGroup,
    model: *mut SmtShardedModel,
    corpus: *mut SmtShardedCorpus,
    tokenizer: Tokenizer,
    unk_id: Option<u32>,
    median_token_len: usize,
}

/// `Document` without the embeddings: they are corpus rows.
pub struct GpuDocument {
    pub filename: String,
    pub lines: Vec<String>,
    pub first_row: u64,
}

impl GpuSearch {
    /// `table`: the model's `embeddings` tensor [vocab x 256] as f32 (what `StaticModel::from_pretrained` loads,
    /// call sites src/cmds/search.rs:123-128); it is uploaded once.
    /// `devices`: the GPU ordinals to use, e.g. from `SEMTOOLS_DEVICES=0,1,2,3` (one entry = today's single GPU).
    pub fn new(devices: &[i32], table: &[f32], vocab: usize, normalize: bool, tokenizer: Tokenizer, unk_id: Option<u32>,
               median_token_len: usize) -> Result<Self> {
        let (mut ctx, mut group, mut model, mut corpus) = (ptr::null_mut(), ptr::null_mut(), ptr::null_mut(), ptr::null_mut());
        unsafe {
            if devices.len() == 1 {
                check(smt_ctx_create(devices[0], &mut ctx))?;
                check(smt_group_from_ctx(ctx, &mut group))?;                 // no RCCL, nothing exchanged
            } else {
                check(smt_group_create(devices.as_ptr(), devices.len() as i32, &mut group))?;   // ncclCommInitAll
            }
            check(smt_sharded_model_create(group, table.as_ptr(), vocab as u64, SMT_DIM, normalize as i32, &mut model))?;
            check(smt_sharded_corpus_create(group, SMT_DIM, &mut corpus))?;
        }
        Ok(Self { ctx, group, model, corpus, tokenizer, unk_id, median_token_len })
    }

    /// model2vec-rs' front half of `encode_with_args`: char pre-truncation, `encode_batch_fast(.., false)`, unk
    /// ids dropped.  Returns the CSR token stream the GPU pools.
    fn tokenize(&self, texts: &[String], max_tokens: usize) -> Result<(Vec<u32>, Vec<u64>)> {
        let budget = max_tokens.saturating_mul(self.median_token_len);
        let cut: Vec<&str> = texts.iter().map(|t| match t.char_indices().nth(budget) {
            Some((i, _)) => &t[..i],
            None => t.as_str(),
        }).collect();
        let enc = self.tokenizer.encode_batch_fast(cut, false).map_err(|e| anyhow::anyhow!(e.to_string()))?;
        let mut ids = Vec::new();
        let mut offsets = vec![0u64];
        for e in &enc {
            ids.extend(e.get_ids().iter().copied().filter(|&i| Some(i) != self.unk_id));
            offsets.push(ids.len() as u64);
        }
        Ok((ids, offsets))
    }

    /// `StaticModel::encode_single` (mod.rs:138,153; src/cmds/search.rs:136): 512-token cap, one row to the host.
    pub fn encode_single(&self, query: &str) -> Result<Vec<f32>> {
        let (ids, offsets) = self.tokenize(&[query.to_string()], 512)?;
        let mut out = vec![0f32; SMT_DIM as usize];
        check(unsafe { smt_sharded_embed(self.model, ids.as_ptr(), offsets.as_ptr(), 1, 512, out.as_mut_ptr(), ptr::null_mut(),
                                         ptr::null_mut()) })?;
        Ok(out)
    }

    /// `create_document_from_content` (mod.rs:49-75): the lines are embedded straight INTO the corpus.
    pub fn create_document_from_content(&mut self, filename: String, content: &str, ignore_case: bool)
        -> Result<Option<GpuDocument>> {
        let lines: Vec<String> = content.lines().map(str::to_string).collect();
        if lines.is_empty() {
            return Ok(None);
        }
        let for_embedding: Vec<String> =
            if ignore_case { lines.iter().map(|s| s.to_lowercase()).collect() } else { lines.clone() };
        let mut first_row = 0u64;
        for batch in for_embedding.chunks(16384) {            // encode_with_args(.., Some(2048), 16384)
            let (ids, offsets) = self.tokenize(batch, 2048)?;
            let mut first = 0u64;
            // lines are dealt to the GPUs in contiguous blocks, pooled by all of them at once and appended to their
            // shards; `first` is the first of the new GLOBAL rows (global row == line order)
            check(unsafe { smt_sharded_embed(self.model, ids.as_ptr(), offsets.as_ptr(), batch.len() as u64, 2048,
                                             ptr::null_mut(), self.corpus, &mut first) })?;
            if batch.as_ptr() == for_embedding.as_ptr() {
                first_row = first;
            }
        }
        Ok(Some(GpuDocument { filename, lines, first_row }))
    }

    /// `search_documents` (mod.rs:77-120).  `documents` must be the documents embedded into this corpus, in order.
    pub fn search_documents(&self, documents: &[GpuDocument], query_embedding: &[f32], config: &SearchConfig)
        -> Result<Vec<SearchResult>> {
        let total = unsafe { smt_sharded_corpus_rows(self.corpus) };
        if total == 0 {
            return Ok(Vec::new());
        }
        // threshold given => every hit comes back and top_k is ignored (mod.rs:115-116); start with a modest buffer
        // and let SMT_E_TRUNCATED report the true count
        let mut cap = if config.max_distance.is_some() { 1024.min(total) } else { config.top_k as u64 };
        let (mut rows, mut dist, mut n) = (Vec::new(), Vec::new(), 0u64);
        loop {
            rows.resize(cap as usize, 0u64);
            dist.resize(cap as usize, 0f64);
            // per-GPU scan + select, ONE all-gather of the per-shard top-k lists, merge: rows come back global
            let rc = unsafe { smt_sharded_search(self.corpus, query_embedding.as_ptr(), 1, config.top_k as u32,
                                                 config.max_distance.unwrap_or(f64::NAN), SMT_MODE_DOCUMENTS, ptr::null(), 0,
                                                 rows.as_mut_ptr(), dist.as_mut_ptr(), &mut n, cap) };
            if rc == SMT_E_TRUNCATED {
                cap = n;
                continue;
            }
            check(rc)?;
            break;
        }
        // rows are already in the reference's order: distance ascending, ties in (document, line) order
        let mut out = Vec::with_capacity(n as usize);
        for i in 0..n as usize {
            let d = documents.partition_point(|doc| doc.first_row + doc.lines.len() as u64 <= rows[i]);
            let doc = &documents[d];
            let idx = (rows[i] - doc.first_row) as usize;
            let bottom = idx.saturating_sub(config.n_lines);                 // mod.rs:90
            let top = min(doc.lines.len(), idx + config.n_lines + 1);        // mod.rs:91
            out.push(SearchResult { filename: doc.filename.clone(), lines: doc.lines[bottom..top].to_vec(),
                                    distance: dist[i], start: bottom, end: top, match_line: idx });
        }
        Ok(out)
    }

    /// `search_files` (mod.rs:122-143): the first unreadable file aborts, as in the reference (`?` at :130).
    pub fn search_files(&mut self, files: &[String], query: &str, config: &SearchConfig) -> Result<Vec<SearchResult>> {
        let mut documents = Vec::new();
        for f in files {
            let content = read_to_string(f)?;
            if let Some(doc) = self.create_document_from_content(f.clone(), &content, config.ignore_case)? {
                documents.push(doc);
            }
        }
        let q = self.encode_single(query)?;
        self.search_documents(&documents, &q, config)
    }

    /// Rows of a document just embedded, for `search_with_workspace` (mod.rs:168-181: one LineEmbedding per line).
    pub fn read_rows(&self, first_row: u64, n: usize) -> Result<Vec<f32>> {
        let mut out = vec![0f32; n * SMT_DIM as usize];
        check(unsafe { smt_sharded_corpus_read_rows(self.corpus, first_row, n as u64, out.as_mut_ptr()) })?;
        Ok(out)
    }

    pub fn group(&self) -> *mut SmtGroup { self.group }
    pub fn model(&self) -> *mut SmtShardedModel { self.model }
}

impl Drop for GpuSearch {
    fn drop(&mut self) {
        unsafe {
            smt_sharded_corpus_destroy(self.corpus);
            smt_sharded_model_destroy(self.model);
            smt_group_destroy(self.group);
            if !self.ctx.is_null() {
                smt_ctx_destroy(self.ctx);                              // (after the group that borrowed it)
            }
        }
    }
}

/// `search_with_workspace` (mod.rs:146-216) with the store of `crate::workspace::hip_store`: new / changed
/// documents are embedded straight into the workspace corpus (no `Vec<Vec<f32>>`, no per-point JSON), then the
/// path-subset search runs on the GPU.  The stderr progress lines and the order of the steps are the reference's.
#[cfg(feature = "workspace")]
pub fn search_with_workspace(gpu: &mut GpuSearch, files: &[String], query: &str, config: &SearchConfig,
                             workspace_name: Option<&str>) -> Result<Vec<crate::workspace::store::RankedLine>> {
    use crate::workspace::hip_store::HipLineStore;
    use crate::workspace::store::{DocumentState, Store};
    use crate::workspace::Workspace;

    let query_embedding = gpu.encode_single(query)?;
    let ws = Workspace::open(workspace_name)?;
    let store = Store::open(&ws.config.root_dir)?;                       // document metadata stays where it is
    let mut lines = HipLineStore::open(gpu.group(), &ws.config.root_dir)?; // replaces line_embeddings.qdrant
    let doc_states = store.analyze_document_states(files)?;              // mod.rs:158 (store.rs:549-611)
    let (mut n_lines, mut docs_to_upsert) = (0usize, Vec::new());
    for state in &doc_states {
        if let DocumentState::Changed(info) | DocumentState::New(info) = state {
            let text = if config.ignore_case { info.content.to_lowercase() } else { info.content.clone() };
            let batch: Vec<String> = text.lines().map(str::to_string).collect();
            if batch.is_empty() {
                continue;                                                // create_document_from_content -> None
            }
            let (ids, offsets) = gpu.tokenize(&batch, 2048)?;
            lines.upsert_document(gpu.model(), &info.filename, &ids, &offsets)?;
            n_lines += batch.len();
            docs_to_upsert.push(info.meta.clone());
        }
    }
    if n_lines > 0 {
        eprintln!("Updating workspace with {} lines from new/changed docs...", n_lines);
        lines.flush()?;
    }
    if !docs_to_upsert.is_empty() {
        eprintln!("Updating workspace with {} new/changed documents...", docs_to_upsert.len());
        store.upsert_document_metadata(&docs_to_upsert)?;
    }
    lines.search_line_embeddings(&query_embedding, files, config.top_k, config.max_distance.map(|d| d as f32))
}
