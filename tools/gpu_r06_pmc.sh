#!/usr/bin/env bash
# Round 6 counters, each group in its own rocprofv3 --pmc pass with --kernel-trace only beside it (gpurun refuses anything else):
#   K3 on the final binary (VERDICT r5 "next" 7): gemm_rowreg_kernel<MODE, IMG> at 1000 x 10 M and 200 x 10 M (the f16 x 1 border
#     moved to 129 queries in round 5), from f32 rows and from the operand image: MFMA-busy, wave cycles waiting, LDS conflicts;
#   the IVF ADC kernels after the PQ kind's conflict-free LUT walk and 8-row re-score (VERDICT r5 "next" 4c): LDS conflicts, VALU, fetch.
# Usage (GPU box, repo root): bash tools/gpu_r06_pmc.sh ; summaries land in gpurun_out/r06_k3_pmc_*.json and r06_ivf_pmc_*.json
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
k3() {  # tag, k3_call.py args, counters...
  local tag="$1" args="$2"; shift 2
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc_k3_$tag" -o k3 -- python "$root/tools/k3_call.py" $args --reps 3 > "$out/pmc_k3_$tag.log" 2>&1
  python "$root/tools/summarize_pmc.py" "$out/pmc_k3_$tag" "$out/r06_k3_pmc_$tag.json" "rocprofv3 --pmc $* --kernel-trace -- python tools/k3_call.py $args --reps 3 (10 M random unit rows, top-10, device-resident calls)" > /dev/null
}
for shape in "1000q_image:--nq 1000" "1000q_f32rows:--nq 1000 --no-image" "200q_image:--nq 200" "200q_f32rows:--nq 200 --no-image"; do
  stag="${shape%%:*}"; sargs="${shape#*:}"
  k3 "${stag}_mfma" "$sargs" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16
  k3 "${stag}_wait" "$sargs" SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU
  k3 "${stag}_lds" "$sargs" SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
done
ivf() {  # name, counters...
  local name="$1"; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc_ivf_$name" -o ivf -- python "$root/tools/ivf_adc_probe.py" > "$out/pmc_ivf_$name.log" 2>&1
  python "$root/tools/summarize_pmc.py" "$out/pmc_ivf_$name" "$out/r06_ivf_pmc_$name.json" "rocprofv3 --pmc $* --kernel-trace -- python tools/ivf_adc_probe.py (10 M rows in 20000 topics, nlist 4096, nprobe 8, rerank 128, 1000 queries, per-list PCA codes then global PQ)" > /dev/null
}
ivf fetch FETCH_SIZE GRBM_GUI_ACTIVE
ivf valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
ivf wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS
ivf lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU
find "$out" -name "*.csv" -size +8M -delete
python - <<'PY'
import json, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r06_k3_pmc_*.json") + glob.glob("/root/repo/gpurun_out/r06_ivf_pmc_*.json")):
    try: d = json.load(open(f))
    except Exception as e: print(f, e); continue
    for k, v in d["kernels"].items():
        if "gemm_rowreg_kernel" in k or "ivf_adc" in k:
            print(f.split("/")[-1], k[:56], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a.endswith("_max") or a.endswith("_avg") or a == "dispatches"})
PY
