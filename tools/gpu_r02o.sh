#!/usr/bin/env bash
# A/B: working tree vs ab_old (git worktree of the previous commit), same box; then PMC passes on the 1000-query batch
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"; out="$root/gpurun_out"; mkdir -p "$out"
timeout 600 python -m pytest tests/test_gpu_batched.py -x -q 2>&1 | tail -2
for rep in 1 2; do
  for side in new old; do
    if [ $side = new ]; then tool="$root/tools/bench_small_batch.py"; else tool="$root/ab_old/tools/bench_small_batch.py"; fi
    echo "--- $side"
    timeout 600 python "$tool" --nq 64 128 256 1000 --variants 1 --reps 5 2>&1 | grep -E "^\{" | cut -c1-200
  done
done
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pmc_k3big_$tag" -o k3 -- python "$root/tools/bench_small_batch.py" --nq 1000 --variants 1 --reps 2 > "$out/pmc_k3big_$tag.log" 2>&1
  python "$root/tools/summarize_pmc.py" "$out/pmc_k3big_$tag" "$out/pmc_k3big_$tag.json" "$set" > /dev/null 2>&1
  python - "$out/pmc_k3big_$tag.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d["kernels"].items():
    if "gemm_level" in k:
        print(k[:60], {a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items() if a.endswith("_max") or a.startswith("max_us") or a=="dispatches"})
PY
done
find "$out" -name "*.csv" -size +4M -delete
