#!/usr/bin/env bash
# round 6 evidence (ONE run): the default bench line + detail; per-leg rocprofv3 --kernel-trace --stats runs reduced to one row per
# (kernel, launch shape) by tools/kernel_shapes.py; the IVF ADC counters on this binary (separate --pmc passes); the small-call
# breakdown.  Usage (GPU box, repo root): bash tools/gpu_r06_final.sh [tag]
tag="${1:-r06}"
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
timeout 1200 python bench.py --detail-out "$out/${tag}_bench_detail.json" > "$out/${tag}_bench_line.json" 2> "$out/${tag}_bench.err"; echo "bench rc=$? line bytes=$(tail -1 "$out/${tag}_bench_line.json" | wc -c)"
cd /tmp && export TMPDIR=/tmp
all_off="--no-cpu-baseline --no-c4 --no-secondary --no-embed --no-ivfpq --no-workspace --no-ingest --no-group-issue --no-small-calls"
for leg in c2 c4 c3 small ivfpq; do
  case $leg in
    c2) flags="$all_off --steps 1000 --warmup 100";;
    c4) flags="${all_off/--no-c4/} --steps 20 --warmup 5";;
    c3) flags="${all_off/--no-secondary/} --steps 20 --warmup 5";;
    small) flags="${all_off/--no-small-calls/} --steps 20 --warmup 5";;
    ivfpq) flags="${all_off/--no-ivfpq/} --c5-full-rows 0 --steps 20 --warmup 5";;
  esac
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}_$leg" -o bench -- python "$root/bench.py" $flags --detail-out "$out/${tag}_bench_detail_prof_$leg.json" > "$out/prof_${tag}_$leg.log" 2>&1
  trace=$(find "$out/prof_${tag}_$leg" -name "*kernel_trace.csv" | head -1)
  stats=$(find "$out/prof_${tag}_$leg" -name "*kernel_stats.csv" | head -1)
  [ -n "$stats" ] && cp "$stats" "$out/${tag}_bench_${leg}_kernel_stats.csv"
  [ -n "$trace" ] && python "$root/tools/kernel_shapes.py" "$trace" > "$out/${tag}_bench_${leg}_kernel_shapes.csv" && head -4 "$out/${tag}_bench_${leg}_kernel_shapes.csv" | cut -c1-160
done
ivf() {  # name, counters...
  local name="$1"; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc_ivf_$name" -o ivf -- python "$root/tools/ivf_adc_probe.py" > "$out/pmc_ivf_$name.log" 2>&1
  python "$root/tools/summarize_pmc.py" "$out/pmc_ivf_$name" "$out/${tag}_ivf_pmc_$name.json" "rocprofv3 --pmc $* --kernel-trace -- python tools/ivf_adc_probe.py (10 M rows in 20000 topics, nlist 4096, nprobe 8, rerank 128, 1000 queries, per-list PCA codes then global PQ)" > /dev/null
}
ivf fetch FETCH_SIZE GRBM_GUI_ACTIVE
ivf valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
ivf wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS
ivf lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU
cd "$root"
find "$out" -name "*kernel_trace.csv" -size +1M -delete
find "$out" -name "*counter_collection.csv" -size +8M -delete
timeout 300 python tools/small_call_breakdown.py > "$out/${tag}_small_calls.jsonl" 2> "$out/${tag}_small_calls.err"; echo "small calls rc=$?"
tail -1 "$out/${tag}_bench_line.json" | cut -c1-3000
