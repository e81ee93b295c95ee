#!/usr/bin/env bash
tag="${1:-r01}"
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
python tools/profile_select.py 2>&1 | grep -v amdgpu.ids | tee "$out/select_phases_$tag.log"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_c3_$tag" -o c3 -- python "$root/tools/bench_batched.py" --rows 10000000 --nq 1000 --reps 2 --check 1 > "$out/prof_c3_$tag.log" 2>&1
grep -v "simple_timer\|SQLite" "$out/prof_c3_$tag.log" | tail -3
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d "$out/pmc_c3_$tag" -o c3 -- python "$root/tools/bench_batched.py" --rows 10000000 --nq 1000 --reps 1 --check 1 > "$out/pmc_c3_$tag.log" 2>&1
grep -v "simple_timer\|SQLite" "$out/pmc_c3_$tag.log" | tail -3
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d "$out/pmc_c3u_$tag" -o c3 -- python "$root/tools/bench_batched.py" --rows 10000000 --nq 1000 --reps 1 --check 1 > "$out/pmc_c3u_$tag.log" 2>&1
grep -v "simple_timer\|SQLite" "$out/pmc_c3u_$tag.log" | tail -2
find "$out" -name "*.csv" -size +8M -delete
ls "$out/prof_c3_$tag" "$out/pmc_c3_$tag" "$out/pmc_c3u_$tag"
