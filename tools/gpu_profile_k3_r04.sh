#!/usr/bin/env bash
# Round 4 (bootstrap + quarter-level plan): counters of the K3 default mode at 1000 x 10 M (f16 x 1) and 128 x 10 M (f16 x 2): MFMA-busy cycles, active
# cycles (clock), LDS instructions / bank conflicts / waits, VALU and wait states.  Separate --pmc passes (counter groups).
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  name="$1"; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc_k3_$name" -o k3 -- python "$root/tools/bench_small_batch.py" --nq 128 1000 --reps 2 $K3_PROFILE_ARGS > "$out/pmc_k3_$name.log" 2>&1
  python "$root/tools/summarize_pmc.py" "$out/pmc_k3_$name" "$out/r04_k3_pmc_$name.json" "rocprofv3 --pmc $* --kernel-trace -- python tools/bench_small_batch.py --nq 128 1000 --reps 2 $K3_PROFILE_ARGS (10 M rows; _max = the main level of the 1000-query batch for gemm_rowreg_kernel<2>, of the 128-query batch for <1>)" > /dev/null
}
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
run wait SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU
run issue SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY
find "$out" -name "*.csv" -size +8M -delete
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_k3_pmc_*.json")):
    d = json.load(open(f))
    for k, v in d["kernels"].items():
        if "gemm_rowreg_kernel" in k:
            print(f.split("/")[-1], k[:48], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a.endswith("_max") or a == "dispatches"})
PY
