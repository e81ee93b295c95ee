#!/usr/bin/env bash
# round 4, last refresh after the chunk-wise reduction reached the one-query scan kernel: the default bench line + detail, and the
# rocprofv3 kernel stats of the c2 and c4 legs (the other legs' kernels did not change since tools/gpu_r04_final.sh r04n)
tag="${1:-r04p}"
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
timeout 600 python bench.py --detail-out "$out/${tag}_bench_detail.json" > "$out/${tag}_bench_line.json" 2> "$out/${tag}_bench.err"; echo "bench rc=$? line bytes=$(tail -1 "$out/${tag}_bench_line.json" | wc -c)"
cd /tmp && export TMPDIR=/tmp
for leg in c2 c4; do
  extra="--no-c4"; steps="--steps 1000 --warmup 100"; [ "$leg" = c4 ] && { extra=""; steps="--steps 20 --warmup 5"; }
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}_$leg" -o bench -- python "$root/bench.py" --no-cpu-baseline $steps $extra --no-secondary --no-embed --no-ivfpq --no-workspace --no-ingest --no-group-issue --detail-out "$out/${tag}_bench_detail_prof_$leg.json" > "$out/prof_${tag}_$leg.log" 2>&1
  cp "$out/prof_${tag}_$leg/bench_kernel_stats.csv" "$out/${tag}_bench_${leg}_kernel_stats.csv" 2>/dev/null
  head -3 "$out/${tag}_bench_${leg}_kernel_stats.csv" | cut -c1-150
done
find "$out" -name "*.csv" -size +8M -delete
