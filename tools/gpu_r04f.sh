#!/usr/bin/env bash
# Round 4, session F: the ratio-4 last level (bootstrap 1/64 -> 1/4 -> 3/4), parity + timelines vs the appended-levels plan; image-scan sweep; ingest.
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
step() { echo "=== $1 ($(date +%T))"; }
step "pytest K3 paths"
timeout 1500 python -m pytest tests/test_gpu_batched.py tests/test_gpu_filtered_batches.py tests/test_gpu_image.py tests/test_gpu_nearties.py tests/test_gpu_defaults.py tests/test_gpu_fullsize.py tests/test_gpu_group.py tests/test_gpu_sharded_store.py -x -q 2>&1 | grep -E "^E |passed|failed|Error" | head -20
step "timelines (bootstrap plan | appended-levels plan)"
for cfg in "1000_img --nq 1000" "1000_f32 --nq 1000 --no-image" "256_img --nq 256 --reps 5" "32_img --nq 32 --reps 10" "1_img --nq 1 --reps 10" "256_subset_img --nq 256 --reps 5 --subset" "1000_1M_img --nq 1000 --rows 1000000 --reps 5"; do
  tag="${cfg%% *}"; args="${cfg#* }"
  bash tools/trace_call.sh "${tag}_boot" $args | tail -10
  bash tools/trace_call.sh "${tag}_old" $args --tune gemm_bootstrap=0 | tail -1
done
step "image scan sweep"
timeout 300 python tools/sweep_image_scan.py 2>&1 | grep rows
step "bench c3 + workspace + ingest"
timeout 400 python bench.py --steps 200 --warmup 50 --no-c4 --no-embed --no-ivfpq --no-cpu-baseline --detail-out "$out/bench_detail_r04f.json" 2> "$out/bench_r04f.err" | tail -c 3500; grep -v "bench detail" "$out/bench_r04f.err" | tail -3
python - <<'PY'
import json
d = json.load(open("/root/repo/gpurun_out/bench_detail_r04f.json"))
print(json.dumps(d.get("ingest"), indent=0)[:1500])
PY
step "done"
