#!/usr/bin/env bash
# Round 4, session B: the filtered-batch route fixed (tile table), issuing threads, K1 nt probe, kernel timelines of K3 calls.
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
step() { echo "=== $1 ($(date +%T))"; }
step "pytest (changed paths)"
timeout 900 python -m pytest tests/test_gpu_filtered_batches.py tests/test_gpu_sharded_store.py tests/test_gpu_group.py tests/test_gpu_embed.py tests/test_gpu_batched.py -x -q 2>&1 | tail -12
step "bench workspace leg"
timeout 300 python bench.py --steps 200 --warmup 50 --no-c4 --no-embed --no-ivfpq --no-ingest --no-cpu-baseline --no-secondary --detail-out "$out/bench_detail_r04b_ws.json" 2> "$out/bench_r04b.err" | tail -c 3000; grep -v "bench detail" "$out/bench_r04b.err" | tail -3
python - <<'PY'
import json
d = json.load(open("/root/repo/gpurun_out/bench_detail_r04b_ws.json"))["workspace"]
for k in ("one_query", "batch", "one_query_image", "batch_image", "checks"):
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in d[k].items() if not isinstance(b, dict)})
PY
step "K1 A/B (1 = r03, 3 = ids prefetched, 11 = + nt row loads)"
timeout 400 python tools/embed_ab.py 1 3 11 > "$out/embed_ab_r04b.json" 2> "$out/embed_ab_r04b.err"; python - <<'PY'
import json
d = json.load(open("/root/repo/gpurun_out/embed_ab_r04b.json"))
for k, v in d.items():
    if k.startswith("embed_batched"): print(k, [(x["zipf_ms"], x["uniform_ms"], x["bit_exact"]) for x in v])
PY
step "group issue"
timeout 300 python tools/bench_group_issue.py > "$out/r04_group_issue_b.json" 2> "$out/group_issue_b.err"; python - <<'PY'
import json
d = json.load(open("/root/repo/gpurun_out/r04_group_issue_b.json"))
for k, v in d.items():
    if k.endswith("_shards"): print(k, json.dumps(v))
print(d.get("verdict"))
PY
tail -3 "$out/group_issue_b.err"
step "timelines"
bash tools/trace_call.sh 1000_img --nq 1000
bash tools/trace_call.sh 1000_f32 --nq 1000 --no-image
bash tools/trace_call.sh 1_img --nq 1 --reps 10
bash tools/trace_call.sh 8_img --nq 8 --reps 10
bash tools/trace_call.sh 256_img --nq 256 --reps 5
bash tools/trace_call.sh 256_subset_img --nq 256 --reps 5 --subset
step "done"
