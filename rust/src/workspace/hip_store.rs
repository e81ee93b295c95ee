//! GPU-resident replacement for the `line_embeddings.qdrant` shard of `src/workspace/store.rs` (reference v3.0.0):
//! `upsert_line_embeddings` (:402-434), `search_line_embeddings` (:481-546), `delete_line_embeddings`,
//! `count_line_embeddings`.  UNCOMPILED here -- see rust/README.md.
//!
//! Storage: one flat corpus file `line_embeddings.f32` (smt_corpus_save format, rows in global = insertion order
//! whatever the number of GPUs that wrote it) + `line_rows.json` (path -> first_row, n_lines; and how the rows were
//! dealt over the GPUs, so that a group of the same size gets the same shards back).  The matrix lives in an
//! `smt_sharded_corpus` on the caller's `smt_group`: every upsert is pooled by all GPUs at once and its rows are dealt
//! over them; every search is a per-GPU scan + one all-gather + merge.  With a one-GPU group these are the single-GPU
//! calls.  A document's lines are contiguous GLOBAL rows, so the reference's payload filter
//! `path IN subset` (:507-515) becomes a sorted list of row ranges, and a re-embedded document simply gets a fresh
//! extent (the reference's stale-tail-rows quirk of upsert-only storage is not reproduced).
use crate::search::hip_ffi::*;
use crate::workspace::store::RankedLine;
use anyhow::Result;
use std::collections::BTreeMap;
use std::ffi::CString;
use std::path::{Path, PathBuf};
use std::ptr;

#[derive(Clone, Copy)]
struct Extent {
    first_row: u64,
    n_lines: u64,
}

/// `line_rows.json` AS THE C++ HOST LAYER WRITES IT (semtools_amd/csrc/host/store.cpp `flush_line_embeddings`): the two
/// implementations open each other's workspaces.
/// `{"extents": [{"path", "first_row", "n_rows"}], "generation": n, "shards": {"n_ranks": n, "pieces": [[rows, rank], ...]}}`
#[derive(serde::Serialize, serde::Deserialize)]
struct ExtentEntry {
    path: String,
    first_row: u64,
    n_rows: u64,
}
#[derive(Default, serde::Serialize, serde::Deserialize)]
struct Shards {
    #[serde(default)]
    n_ranks: i32,
    /// (rows, rank) per piece of the global row numbering, in global order
    #[serde(default)]
    pieces: Vec<(u64, u32)>,
}
/// `deny_unknown_fields`: every field here has a default, so without it ANY JSON object -- a round-2 table, whose keys are paths --
/// would parse as an empty `Current` table before the older forms are tried: the store would re-embed every document
/// and leave the old rows behind as garbage.
#[derive(Default, serde::Serialize, serde::Deserialize)]
#[serde(deny_unknown_fields)]
struct RowsFile {
    #[serde(default)]
    extents: Vec<ExtentEntry>,
    /// bumped whenever rows move or are rewritten in place; `line_index.gen` must name the same value for an index to be loaded
    #[serde(default)]
    generation: u64,
    #[serde(default)]
    shards: Shards,
}

/// What earlier revisions of THIS file wrote (round 2: a bare map path -> extent; round 3: `{extents: {path: ..}, n_ranks, pieces}`).
/// Read for compatibility, never written.
#[derive(serde::Deserialize)]
struct OldExtent {
    first_row: u64,
    n_lines: u64,
}
#[derive(serde::Deserialize)]
struct OldRowsFile {
    extents: BTreeMap<String, OldExtent>,
    #[serde(default)]
    n_ranks: i32,
    #[serde(default)]
    pieces: Vec<(u64, u32)>,
}
#[derive(serde::Deserialize)]
#[serde(untagged)]
enum AnyRowsFile {
    Current(RowsFile),
    Round3(OldRowsFile),
    Round2(BTreeMap<String, OldExtent>),
}

/// (extents, generation, n_ranks, pieces) of whatever form the file has; an unreadable table is an EMPTY one -- every document then
/// counts as changed and is embedded again (a parse error must not brick the workspace)
fn read_rows_file(text: &str) -> (BTreeMap<String, Extent>, u64, i32, Vec<(u64, u32)>) {
    match serde_json::from_str::<AnyRowsFile>(text) {
        Ok(AnyRowsFile::Current(f)) => (
            f.extents.into_iter().map(|e| (e.path, Extent { first_row: e.first_row, n_lines: e.n_rows })).collect(),
            f.generation, f.shards.n_ranks, f.shards.pieces),
        Ok(AnyRowsFile::Round3(f)) => (
            f.extents.into_iter().map(|(p, e)| (p, Extent { first_row: e.first_row, n_lines: e.n_lines })).collect(), 0, f.n_ranks, f.pieces),
        Ok(AnyRowsFile::Round2(m)) => (
            m.into_iter().map(|(p, e)| (p, Extent { first_row: e.first_row, n_lines: e.n_lines })).collect(), 0, 0, Vec::new()),
        Err(_) => (BTreeMap::new(), 0, 0, Vec::new()),
    }
}

pub struct HipLineStore {
    group: *mut SmtGroup,
    corpus: *mut SmtShardedCorpus,
    extents: BTreeMap<String, Extent>,
    dir: PathBuf,
    rows_on_disk: u64,
    /// rows [rows_on_disk, rows_written_ahead) already sit in the file (queued by `write_rows_ahead`; not durable, not named by the
    /// header until `flush` commits)
    rows_written_ahead: u64,
    generation: u64,
}

impl HipLineStore {
    pub fn open(group: *mut SmtGroup, root_dir: &str) -> Result<Self> {
        let dir = Path::new(root_dir).to_path_buf();
        let file = dir.join("line_embeddings.f32");
        let mut corpus = ptr::null_mut();
        let (mut extents, mut generation, mut file_ranks, mut pieces) = (BTreeMap::new(), 0u64, 0i32, Vec::new());
        if file.exists() {
            let c = CString::new(file.to_string_lossy().as_bytes())?;
            if let Ok(text) = std::fs::read_to_string(dir.join("line_rows.json")) {
                (extents, generation, file_ranks, pieces) = read_rows_file(&text);
            }
            let mut n_ranks = 0i32;
            check(unsafe { smt_group_info(group, &mut n_ranks, ptr::null_mut(), ptr::null_mut(), ptr::null_mut(), ptr::null_mut()) })?;
            // the same number of GPUs as the run that wrote the store: every shard gets its rows back; otherwise the
            // matrix is cut into ceil(N / n_ranks) ranges (the file is in global row order either way)
            let mut rc = SMT_E_INVALID;
            if n_ranks > 1 && file_ranks == n_ranks && !pieces.is_empty() {
                let rows: Vec<u64> = pieces.iter().map(|p| p.0).collect();
                let rank: Vec<u32> = pieces.iter().map(|p| p.1).collect();
                rc = unsafe { smt_sharded_corpus_load_layout(group, c.as_ptr(), rows.as_ptr(), rank.as_ptr(), rows.len() as u64, &mut corpus) };
            }
            if rc == SMT_E_INVALID {
                rc = unsafe { smt_sharded_corpus_load(group, c.as_ptr(), &mut corpus) };
            }
            check(rc)?;
        } else {
            check(unsafe { smt_sharded_corpus_create(group, SMT_DIM, &mut corpus) })?;
        }
        let rows_on_disk = if file.exists() { unsafe { smt_sharded_corpus_rows(corpus) } } else { 0 };
        // torn write: an extent that points past the rows on disk is dropped (its document is embedded again)
        extents.retain(|_, e: &mut Extent| e.first_row + e.n_lines <= rows_on_disk);
        Ok(Self { group, corpus, extents, dir, rows_on_disk, rows_written_ahead: 0, generation })
    }

    /// `upsert_line_embeddings` for one document: its lines are pooled on the GPUs straight into fresh corpus rows
    /// (`first` = the first of the new global rows; the document's lines stay contiguous in the global numbering).
    pub fn upsert_document(&mut self, model: *mut SmtShardedModel, path: &str, ids: &[u32], offsets: &[u64]) -> Result<()> {
        let n = (offsets.len() - 1) as u64;
        let mut first = 0u64;
        check(unsafe { smt_sharded_embed(model, ids.as_ptr(), offsets.as_ptr(), n, 2048, ptr::null_mut(), self.corpus, &mut first) })?;
        self.extents.insert(path.to_string(), Extent { first_row: first, n_lines: n });
        self.write_rows_ahead();
        Ok(())
    }

    /// What `Store::write_rows_ahead` of the C++ host layer does (semtools_amd/csrc/host/store.cpp; the reference flushes every
    /// 1000-point chunk while it goes, src/workspace/store.rs:402-434): the rows embedded so far are queued to the library's
    /// background file writer while the next document is tokenised and pooled.  Nothing is durable and the header is untouched
    /// until `flush`; any failure here (several shards: SMT_E_UNSUPPORTED) just leaves everything for the commit.
    fn write_rows_ahead(&mut self) {
        let path = self.dir.join("line_embeddings.f32");
        let exists = path.exists();
        if !exists && self.rows_on_disk != 0 { return; }
        let Ok(file) = CString::new(path.to_string_lossy().as_bytes()) else { return };
        let rows = unsafe { smt_sharded_corpus_rows(self.corpus) };
        let written = self.rows_on_disk.max(self.rows_written_ahead);
        if rows <= written { return; }
        let flags = SMT_APPEND_WRITE_AHEAD | if exists { 0 } else { SMT_APPEND_CREATE };
        if unsafe { smt_sharded_corpus_append_to_file_ex(self.corpus, file.as_ptr(), self.rows_on_disk, written, flags) } == SMT_OK {
            self.rows_written_ahead = rows;
        }
    }

    /// `flush_line_embeddings`: append the new rows to the file (O(new rows): every GPU writes its own pieces), then
    /// the row table with the layout.
    pub fn flush(&mut self) -> Result<()> {
        let file = CString::new(self.dir.join("line_embeddings.f32").to_string_lossy().as_bytes())?;
        let rows = unsafe { smt_sharded_corpus_rows(self.corpus) };
        if self.rows_on_disk == 0 && self.rows_written_ahead == 0 {
            check(unsafe { smt_sharded_corpus_save(self.corpus, file.as_ptr()) })?;
        } else {
            // the commit: what was not written ahead goes out now, fsync, header last
            let written = rows.min(self.rows_on_disk.max(self.rows_written_ahead));
            check(unsafe { smt_sharded_corpus_append_to_file_ex(self.corpus, file.as_ptr(), self.rows_on_disk, written, 0) })?;
        }
        self.rows_on_disk = rows;
        self.rows_written_ahead = 0;
        let mut n_ranks = 0i32;
        check(unsafe { smt_group_info(self.group, &mut n_ranks, ptr::null_mut(), ptr::null_mut(), ptr::null_mut(), ptr::null_mut()) })?;
        let n = unsafe { smt_sharded_corpus_layout(self.corpus, ptr::null_mut(), ptr::null_mut(), 0) } as usize;
        let (mut rows, mut rank) = (vec![0u64; n], vec![0u32; n]);
        unsafe { smt_sharded_corpus_layout(self.corpus, rows.as_mut_ptr(), rank.as_mut_ptr(), n as u64) };
        let table = RowsFile {
            extents: self.extents.iter().map(|(p, e)| ExtentEntry { path: p.clone(), first_row: e.first_row, n_rows: e.n_lines }).collect(),
            generation: self.generation,
            shards: if n_ranks > 1 { Shards { n_ranks, pieces: rows.into_iter().zip(rank).collect() } } else { Shards::default() },
        };
        // (a sibling first, then rename: the table is replaced atomically, as store.cpp's write_file_atomic does)
        let tmp = self.dir.join("line_rows.json.tmp");
        std::fs::write(&tmp, serde_json::to_string_pretty(&table)?)?;
        std::fs::rename(&tmp, self.dir.join("line_rows.json"))?;
        Ok(())
    }

    pub fn delete_line_embeddings(&mut self, paths: &[String]) {
        for p in paths {
            self.extents.remove(p);
        }
    }

    pub fn count_line_embeddings(&self) -> u64 {
        self.extents.values().map(|e| e.n_lines).sum()
    }

    /// `Store::search_line_embeddings` (:481-546): rows with score > 1 - max_distance (f32), then ALWAYS the best
    /// top_k (:543); empty subset or top_k == 0 -> [] (:489-491).
    pub fn search_line_embeddings(&self, query_vec: &[f32], subset_paths: &[String], top_k: usize,
                                  max_distance: Option<f32>) -> Result<Vec<RankedLine>> {
        if subset_paths.is_empty() || top_k == 0 {
            return Ok(Vec::new());
        }
        let mut owners: Vec<(SmtRange, &str)> = subset_paths.iter()
            .filter_map(|p| self.extents.get(p).map(|e| (SmtRange { begin: e.first_row, end: e.first_row + e.n_lines }, p.as_str())))
            .collect();
        owners.sort_by_key(|(r, _)| r.begin);
        owners.dedup_by_key(|(r, _)| r.begin);
        if owners.is_empty() {
            return Ok(Vec::new());
        }
        let ranges: Vec<SmtRange> = owners.iter().map(|(r, _)| *r).collect();
        let (mut rows, mut dist, mut n) = (vec![0u64; top_k], vec![0f64; top_k], 0u64);
        // global ranges in, global rows out: the library cuts the ranges along the shards' pieces
        check(unsafe { smt_sharded_search(self.corpus, query_vec.as_ptr(), 1, top_k as u32,
                                          max_distance.map(|d| d as f64).unwrap_or(f64::NAN), SMT_MODE_WORKSPACE,
                                          ranges.as_ptr(), ranges.len() as u32, rows.as_mut_ptr(), dist.as_mut_ptr(), &mut n,
                                          top_k as u64) })?;
        Ok((0..n as usize).map(|i| {
            let j = owners.partition_point(|(r, _)| r.end <= rows[i]);
            let (r, path) = owners[j];
            RankedLine { path: path.to_string(), line_number: (rows[i] - r.begin) as i32, distance: dist[i] as f32 }  // :527-532
        }).collect())
    }
}

impl Drop for HipLineStore {
    fn drop(&mut self) {
        unsafe { smt_sharded_corpus_destroy(self.corpus) };
    }
}

#[cfg(test)]
mod rows_file_formats {
    use super::read_rows_file;

    #[test]
    fn current_form() {
        let t = r#"{"extents":[{"path":"a.txt","first_row":0,"n_rows":3},{"path":"b.txt","first_row":3,"n_rows":2}],"generation":7,
                    "shards":{"n_ranks":2,"pieces":[[3,0],[2,1]]}}"#;
        let (ext, gen, ranks, pieces) = read_rows_file(t);
        assert_eq!(ext.len(), 2);
        assert_eq!(ext["b.txt"].first_row, 3);
        assert_eq!((gen, ranks, pieces.len()), (7, 2, 2));
    }

    #[test]
    fn round3_form() {
        let t = r#"{"extents":{"a.txt":{"first_row":0,"n_lines":3}},"n_ranks":1,"pieces":[[3,0]]}"#;
        let (ext, gen, ranks, pieces) = read_rows_file(t);
        assert_eq!(ext["a.txt"].n_lines, 3);
        assert_eq!((gen, ranks, pieces.len()), (0, 1, 1));
    }

    #[test]
    fn round2_form_is_not_mistaken_for_an_empty_current_table() {
        let t = r#"{"a.txt":{"first_row":0,"n_lines":3},"dir/b.txt":{"first_row":3,"n_lines":9}}"#;
        let (ext, gen, ranks, pieces) = read_rows_file(t);
        assert_eq!(ext.len(), 2);
        assert_eq!(ext["dir/b.txt"].first_row, 3);
        assert_eq!((gen, ranks, pieces.len()), (0, 0, 0));
    }

    #[test]
    fn garbage_is_an_empty_table() {
        assert!(read_rows_file("not json").0.is_empty());
    }
}
