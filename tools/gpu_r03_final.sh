#!/usr/bin/env bash
# Round 3 closing run: the whole GPU suite, smoke(), the default bench line, rocprofv3 kernel stats and FETCH_SIZE of the same line.
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
timeout 1500 python -m pytest tests -q -m gpu -x > "$out/r03_gpu_suite.log" 2>&1; echo "suite rc=$?" ; tail -3 "$out/r03_gpu_suite.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > "$out/bench_r03_final.json" 2> "$out/bench_r03_final.err"; echo "bench rc=$?"; tail -c 600 "$out/bench_r03_final.json" | head -c 300; echo
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_r03f" -o bench -- python "$root/bench.py" --steps 400 --warmup 50 --no-cpu-baseline > "$out/prof_r03f.log" 2>&1
cp "$out/prof_r03f/bench_kernel_stats.csv" "$out/r03_final_bench_kernel_stats.csv" 2>/dev/null
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch_r03f" -o bench -- python "$root/bench.py" --steps 20 --warmup 3 --settle-steps 8 --no-cpu-baseline --no-ivfpq > "$out/pmc_fetch_r03f.log" 2>&1
python "$root/tools/summarize_pmc.py" "$out/pmc_fetch_r03f" "$out/r03_final_pmc_fetch.json" "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 20 --warmup 3 --settle-steps 8 --no-cpu-baseline --no-ivfpq" > /dev/null 2>&1
find "$out" -name "*.csv" -size +8M -delete
ls -la "$out" | grep "r03_final\|bench_r03_final\|r03_gpu_suite"
