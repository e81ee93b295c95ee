#!/usr/bin/env bash
# Round 4, session D: bootstrap plan with the bitwise tau kernel (parity + timelines), IVF ADC counters.
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
step() { echo "=== $1 ($(date +%T))"; }
step "pytest K3 paths"
timeout 1200 python -m pytest tests/test_gpu_batched.py tests/test_gpu_filtered_batches.py tests/test_gpu_image.py tests/test_gpu_nearties.py tests/test_gpu_defaults.py -x -q 2>&1 | grep -E "^E |passed|failed|Error" | head -20
step "timelines (bootstrap plan | appended-levels plan)"
for cfg in "1000_img --nq 1000" "1000_f32 --nq 1000 --no-image" "256_img --nq 256 --reps 5" "32_img --nq 32 --reps 10" "1_img --nq 1 --reps 10" "256_subset_img --nq 256 --reps 5 --subset"; do
  tag="${cfg%% *}"; args="${cfg#* }"
  bash tools/trace_call.sh "${tag}_boot" $args | tail -11
  bash tools/trace_call.sh "${tag}_old" $args --tune gemm_bootstrap=0 | tail -1
done
step "IVF ADC counters"
cd /tmp && export TMPDIR=/tmp
pmc() {  # name, counters...
  name="$1"; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc_ivf_$name" -o ivf -- python "$root/tools/ivf_adc_probe.py" > "$out/pmc_ivf_$name.log" 2>&1
  python "$root/tools/summarize_pmc.py" "$out/pmc_ivf_$name" "$out/r04_ivf_pmc_$name.json" "rocprofv3 --pmc $* --kernel-trace -- python tools/ivf_adc_probe.py (10 M rows in 20000 topics, nlist 4096, nprobe 8, rerank 128, 1000 queries, per-list PCA codes then global PQ)" > /dev/null
}
pmc fetch FETCH_SIZE GRBM_GUI_ACTIVE
pmc valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pmc wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU
find "$out" -name "*.csv" -size +8M -delete
python - <<'PY'
import json, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04_ivf_pmc_*.json")):
    d = json.load(open(f))
    for k, v in d["kernels"].items():
        if "ivf_adc" in k: print(f.split("/")[-1], k[:50], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a.endswith("_avg") or a in ("dispatches", "avg_us_under_pmc")})
PY
step "done"
