#!/usr/bin/env bash
# round-2 rocprofv3 evidence: kernel stats + FETCH_SIZE for the c2 bench and for the K3-small kernel
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --no-ivfpq --no-c4"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_r02_c2" -o bench -- python "$root/bench.py" --steps 1000 --warmup 100 $B > "$out/prof_r02_c2.log" 2>&1
grep '"metric"' "$out/prof_r02_c2.log" | cut -c1-300
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_r02_c2" -o bench -- python "$root/bench.py" --steps 20 --warmup 3 --settle-steps 8 $B > "$out/pmc_r02_c2.log" 2>&1
python "$root/tools/summarize_pmc.py" "$out/pmc_r02_c2" "$out/r02_traffic_c2.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 20 --warmup 3 --settle-steps 8 $B" | cut -c1-600
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_r02_k3s" -o k3s -- python "$root/tools/bench_small_batch.py" --nq 32 64 128 --variants 1 --reps 5 > "$out/prof_r02_k3s.log" 2>&1
grep '"nq"' "$out/prof_r02_k3s.log" | cut -c1-330
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_r02_k3s" -o k3s -- python "$root/tools/bench_small_batch.py" --nq 32 64 --variants 1 --reps 2 > "$out/pmc_r02_k3s.log" 2>&1
python "$root/tools/summarize_pmc.py" "$out/pmc_r02_k3s" "$out/r02_traffic_k3s.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/bench_small_batch.py --nq 32 64 --variants 1 --reps 2 (10 M rows: 10.24 GB algorithmic per corpus pass; the main level reads 15/16 of it)" | cut -c1-900
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d "$out/pmc_r02_k3s_mfma" -o k3s -- python "$root/tools/bench_small_batch.py" --nq 32 64 128 --variants 1 --reps 2 > "$out/pmc_r02_k3s_mfma.log" 2>&1
python "$root/tools/summarize_pmc.py" "$out/pmc_r02_k3s_mfma" "$out/r02_mfma_k3s.json" "rocprofv3 --pmc MfmaUtil --kernel-trace -- python tools/bench_small_batch.py --nq 32 64 128 --variants 1 --reps 2" | cut -c1-900
find "$out" -name "*kernel_stats.csv" | head
find "$out" -name "*.csv" -size +6M -delete
