#!/usr/bin/env bash
# One GPU-box session: parity tests, smoke, tuning sweep, bench, rocprofv3 traces.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag>
tag="${1:-r01}"
root="$(pwd)"
out="$root/gpurun_out"
mkdir -p "$out"
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > "$out/pytest_gpu_$tag.log"; cat "$out/pytest_gpu_$tag.log"
SEMTOOLS_NO_TORCH_PRELOAD=1 python -c "import ctypes; L=ctypes.CDLL('semtools_amd/lib/libsemtools_hip.so'); print('standalone system-HIP device count:', L.smt_device_count())" > "$out/standalone_$tag.log" 2>&1; cat "$out/standalone_$tag.log"
python __graft_entry__.py smoke > "$out/smoke_$tag.log" 2>&1; tail -3 "$out/smoke_$tag.log"
timeout 900 python bench.py > "$out/bench_$tag.json" 2> "$out/bench_$tag.err"; cat "$out/bench_$tag.json"; tail -5 "$out/bench_$tag.err"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$tag" -o bench -- python "$root/bench.py" --steps 1000 --warmup 100 --no-cpu-baseline --no-secondary --no-ivfpq > "$out/prof_$tag.log" 2>&1
tail -3 "$out/prof_$tag.log"
find "$out/prof_$tag" -name "*stats*" | head
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch_$tag" -o bench -- python "$root/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-ivfpq > "$out/pmc_fetch_$tag.log" 2>&1
tail -3 "$out/pmc_fetch_$tag.log"
# keep only the small summaries (traces can be large)
find "$out" -name "*.csv" -size +8M -delete
ls -la "$out" "$out/prof_$tag" "$out/pmc_fetch_$tag" 2>/dev/null | head -40
