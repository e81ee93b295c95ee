#!/usr/bin/env bash
# round 5 counters (separate rocprofv3 --pmc runs, --kernel-trace only beside them): the IVF ADC kernels after their split by coding
# kind (ivf_adc_kernel<256, 1> per-list PCA / <256, 0> global PQ), and a FETCH_SIZE pass over the bench for profiles/traffic.json.
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pmc() {  # name, counters...
  name="$1"; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc_ivf_$name" -o ivf -- python "$root/tools/ivf_adc_probe.py" > "$out/pmc_ivf_$name.log" 2>&1
  python "$root/tools/summarize_pmc.py" "$out/pmc_ivf_$name" "$out/r05_ivf_pmc_$name.json" "rocprofv3 --pmc $* --kernel-trace -- python tools/ivf_adc_probe.py (10 M rows in 20000 topics, nlist 4096, nprobe 8, rerank 128, 1000 queries, per-list PCA codes then global PQ)" > /dev/null
}
pmc fetch FETCH_SIZE GRBM_GUI_ACTIVE
pmc valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pmc wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU
cmd="--steps 20 --warmup 3 --settle-steps 8 --no-cpu-baseline --no-ivfpq --no-ingest --no-group-issue --c4-steps 3"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch_r05" -o bench -- python "$root/bench.py" $cmd --detail-out "$out/r05_bench_detail_pmc.json" > "$out/pmc_fetch_r05.log" 2>&1
python "$root/tools/summarize_pmc.py" "$out/pmc_fetch_r05" "$out/r05_pmc_fetch.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py $cmd" > /dev/null 2>&1
find "$out" -name "*.csv" -size +8M -delete
python - <<'PY'
import json, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r05_ivf_pmc_*.json")) + ["/root/repo/gpurun_out/r05_pmc_fetch.json"]:
    try: d = json.load(open(f))
    except Exception as e: print(f, e); continue
    for k, v in d["kernels"].items():
        if "ivf_adc" in k or "pmc_fetch" in f and "FETCH_SIZE_avg" in v:
            print(f.split("/")[-1], k[:60], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a.endswith("_avg") or a in ("dispatches", "avg_us_under_pmc", "FETCH_SIZE_min", "FETCH_SIZE_max")})
PY
