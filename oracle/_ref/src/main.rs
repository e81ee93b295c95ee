//! semtools-pin -- step 2 of the pinning recipe (oracle/_ref/README.md).  TEST INFRASTRUCTURE: it runs the REAL reference
//! (run-llama/semtools v3.0.0 as a library, model2vec-rs 0.1.3, simsimd 6.5.1, qdrant-edge through semtools' Store) on
//! the inputs written by make_inputs.py and stores what they return, raw, under oracle/_ref/out/.  collect.py turns that
//! into tests/golden/ref_*.npz, which tests/test_oracle.py compares the C oracle with.
//!
//! UNCOMPILED in the round's container (no cargo, no network).  Written against the public API of the pinned versions:
//!   model2vec_rs::model::StaticModel::{from_pretrained, encode_with_args, encode_single}   (call sites in the reference:
//!       src/cmds/search.rs:123-128,136,154; src/search/mod.rs:69,138,153)
//!   <f32 as simsimd::SpatialSimilarity>::cosine                                            (src/search/mod.rs:86)
//!   semtools::search::{Document, SearchConfig, search_documents}                           (src/search/mod.rs:18-47,77-120)
//!   semtools::workspace::store::{Store, LineEmbedding}                                     (src/workspace/store.rs:67-73,113,402,481)
//!
//! Usage:  cargo run --release -- <inputs dir> <out dir>
use anyhow::{Context, Result};
use model2vec_rs::model::StaticModel;
use semtools::search::{search_documents, Document, SearchConfig};
use semtools::workspace::store::{LineEmbedding, Store};
use simsimd::SpatialSimilarity;
use std::fs;
use std::path::Path;

const DIM: usize = 256;

fn read_f32(path: &Path) -> Result<Vec<f32>> {
    let bytes = fs::read(path).with_context(|| format!("read {}", path.display()))?;
    Ok(bytes.chunks_exact(4).map(|b| f32::from_le_bytes([b[0], b[1], b[2], b[3]])).collect())
}

fn write_f32(path: &Path, v: &[f32]) -> Result<()> {
    let mut bytes = Vec::with_capacity(v.len() * 4);
    for x in v {
        bytes.extend_from_slice(&x.to_le_bytes());
    }
    fs::write(path, bytes).with_context(|| format!("write {}", path.display()))
}

fn write_f64(path: &Path, v: &[f64]) -> Result<()> {
    let mut bytes = Vec::with_capacity(v.len() * 8);
    for x in v {
        bytes.extend_from_slice(&x.to_le_bytes());
    }
    fs::write(path, bytes).with_context(|| format!("write {}", path.display()))
}

fn flatten(rows: &[Vec<f32>]) -> Vec<f32> {
    rows.iter().flat_map(|r| r.iter().copied()).collect()
}

fn main() -> Result<()> {
    let args: Vec<String> = std::env::args().collect();
    let inputs = Path::new(args.get(1).map(String::as_str).unwrap_or("inputs"));
    let out = Path::new(args.get(2).map(String::as_str).unwrap_or("out"));
    fs::create_dir_all(out)?;

    // ------------------------------------------------------------------ A1/A3/A4: model2vec-rs
    // from_pretrained on a local directory (tokenizer.json, model.safetensors, config.json), exactly as the reference calls it
    let model_dir = inputs.join("model");
    let model = StaticModel::from_pretrained(model_dir.to_str().unwrap(), None, None, None)
        .map_err(|e| anyhow::anyhow!("from_pretrained: {e}"))?;
    let lines: Vec<String> = fs::read_to_string(inputs.join("lines.txt"))?.lines().map(str::to_string).collect();
    // the reference's line call (src/search/mod.rs:69) ...
    let emb = model.encode_with_args(&lines, Some(2048), 16384);
    write_f32(&out.join("embed_lines_2048.f32"), &flatten(&emb))?;
    // ... the same lines under the query cap of encode() / encode_single (512), and a tight cap that cuts most lines
    write_f32(&out.join("embed_lines_512.f32"), &flatten(&model.encode_with_args(&lines, Some(512), 1024)))?;
    write_f32(&out.join("embed_lines_4.f32"), &flatten(&model.encode_with_args(&lines, Some(4), 16384)))?;
    // encode_single (src/cmds/search.rs:136; src/search/mod.rs:138,153)
    let queries: Vec<String> = fs::read_to_string(inputs.join("queries.txt"))?.lines().map(str::to_string).collect();
    let qemb: Vec<Vec<f32>> = queries.iter().map(|q| model.encode_single(q)).collect();
    write_f32(&out.join("embed_queries_single.f32"), &flatten(&qemb))?;

    // ------------------------------------------------------------------ A5: simsimd
    let corpus = read_f32(&inputs.join("corpus.f32"))?;
    let qs = read_f32(&inputs.join("queries.f32"))?;
    let (n_rows, n_q) = (corpus.len() / DIM, qs.len() / DIM);
    let mut cos = Vec::with_capacity(n_rows * n_q);
    for qi in 0..n_q {
        let q = &qs[qi * DIM..(qi + 1) * DIM];
        for r in 0..n_rows {
            // the reference skips a row when this is None (src/search/mod.rs:86): record NaN for that case
            cos.push(f32::cosine(q, &corpus[r * DIM..(r + 1) * DIM]).unwrap_or(f64::NAN));
        }
    }
    write_f64(&out.join("simsimd_cosine_f64.bin"), &cos)?;   // [n_q][n_rows]

    // ------------------------------------------------------------------ A6: the reference's own search_documents
    // two documents (400 + 200 lines) whose embeddings are the corpus rows; the text of a line is its row number
    let split = 400usize;
    let mk = |name: &str, lo: usize, hi: usize| Document {
        filename: name.to_string(),
        lines: (lo..hi).map(|i| format!("row {i}")).collect(),
        embeddings: (lo..hi).map(|i| corpus[i * DIM..(i + 1) * DIM].to_vec()).collect(),
    };
    let docs = vec![mk("doc0", 0, split), mk("doc1", split, n_rows)];
    let mut cases = Vec::new();
    for qi in 0..n_q {
        let q = &qs[qi * DIM..(qi + 1) * DIM];
        let mut run = |n_lines: usize, top_k: usize, max_distance: Option<f64>| {
            let cfg = SearchConfig { n_lines, top_k, max_distance, ignore_case: false };
            let res = search_documents(&docs, q, &cfg);
            let hits: Vec<serde_json::Value> = res.iter().map(|r| serde_json::json!({
                "filename": r.filename, "start": r.start, "end": r.end, "match_line": r.match_line,
                "distance_bits": r.distance.to_bits(), "distance": r.distance, "n_context_lines": r.lines.len(),
            })).collect();
            cases.push(serde_json::json!({"query": qi, "n_lines": n_lines, "top_k": top_k, "max_distance": max_distance, "hits": hits}));
        };
        for k in [1usize, 3, 10] {
            run(3, k, None);
        }
        run(3, 3, Some(0.9));      // threshold given: every hit under it, top_k ignored (mod.rs:115-116)
        run(0, 5, Some(0.0));      // strict `<`: nothing is below 0
    }
    fs::write(out.join("search_documents.json"), serde_json::to_string_pretty(&cases)?)?;

    // ------------------------------------------------------------------ A9/A10: the workspace store (qdrant-edge)
    let store_rows = read_f32(&inputs.join("store_rows.f32"))?;
    let n_store = store_rows.len() / DIM;
    let tmp = tempfile::tempdir()?;
    let store = Store::open(tmp.path().to_str().unwrap())?;
    // rows 0..2 = the reference's own known-answer test (store.rs:814-850): doc1 line 0, doc2 line 0, doc2 line 1;
    // the rest: doc3, lines 0..
    let mut les = Vec::new();
    for r in 0..n_store {
        let (path, line) = match r {
            0 => ("doc1", 0),
            1 => ("doc2", 0),
            2 => ("doc2", 1),
            _ => ("doc3", (r - 3) as i32),
        };
        les.push(LineEmbedding { path: path.to_string(), line_number: line, embedding: store_rows[r * DIM..(r + 1) * DIM].to_vec() });
    }
    store.upsert_line_embeddings(&les)?;
    let mut store_cases = Vec::new();
    let q01 = vec![0.1f32; DIM];
    let all: Vec<String> = ["doc1", "doc2", "doc3"].iter().map(|s| s.to_string()).collect();
    let probes: Vec<(&str, Vec<f32>)> = vec![("q_0.1", q01), ("row10", store_rows[10 * DIM..11 * DIM].to_vec()), ("vecq0", qs[..DIM].to_vec())];
    for (name, q) in &probes {
        for (subset, top_k, max_d) in [
            (vec!["doc1".to_string()], 1usize, Some(0.1f32)),       // the reference's known answer: (doc1, 0, d < 0.1)
            (all.clone(), 5, None),
            (all.clone(), 5, Some(0.5f32)),
            (vec!["doc3".to_string()], 7, None),
            (vec!["doc2".to_string(), "doc3".to_string()], 3, Some(0.95f32)),
        ] {
            let ranked = store.search_line_embeddings(q, &subset, top_k, max_d)?;
            let hits: Vec<serde_json::Value> = ranked.iter().map(|r| serde_json::json!({
                "path": r.path, "line_number": r.line_number, "distance_bits": r.distance.to_bits(), "distance": r.distance,
            })).collect();
            store_cases.push(serde_json::json!({"query": name, "subset": subset, "top_k": top_k, "max_distance": max_d, "hits": hits}));
        }
    }
    fs::write(out.join("store_search.json"), serde_json::to_string_pretty(&store_cases)?)?;
    fs::write(out.join("versions.json"), serde_json::to_string_pretty(&serde_json::json!({
        "semtools": "v3.0.0 (git tag)", "model2vec-rs": "0.1.3", "simsimd": "6.5.1",
        "host": std::env::consts::ARCH,
    }))?)?;
    println!("wrote {}", out.display());
    Ok(())
}
