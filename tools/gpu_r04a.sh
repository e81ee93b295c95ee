#!/usr/bin/env bash
# Round 4, session A: parity tests, the new bench line, K1 A/B + counters, the host-issue cost of an 8-shard group, sharded IVF.
# Every step under its own timeout; outputs under gpurun_out/.
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
step() { echo "=== $1 ($(date +%T))"; }
step "pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > "$out/pytest_gpu_r04a.log"; cat "$out/pytest_gpu_r04a.log"
step "bench"
timeout 600 python bench.py > "$out/bench_r04a.json" 2> "$out/bench_r04a.err"; tail -c 6500 "$out/bench_r04a.json"; grep -v "bench detail" "$out/bench_r04a.err" | tail -5
cp "$out/bench_detail.json" "$out/bench_detail_r04a.json" 2>/dev/null
step "K1 A/B"
timeout 400 python tools/embed_ab.py 1 3 7 > "$out/embed_ab_r04a.json" 2> "$out/embed_ab_r04a.err"; cat "$out/embed_ab_r04a.json"; tail -3 "$out/embed_ab_r04a.err"
step "group issue"
timeout 300 python tools/bench_group_issue.py > "$out/r04_group_issue.json" 2> "$out/group_issue.err"; cat "$out/r04_group_issue.json"; tail -3 "$out/group_issue.err"
step "sharded ivf"
timeout 600 python tools/bench_sharded_ivf.py > "$out/r04_sharded_ivf.json" 2> "$out/sharded_ivf.err"; cat "$out/r04_sharded_ivf.json"; tail -3 "$out/sharded_ivf.err"
step "K1 counters"
cd /tmp && export TMPDIR=/tmp
pmc() {  # name, counters...
  name="$1"; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc_k1_$name" -o k1 -- python "$root/tools/bench_embed.py" --uniform --vocab 4000000 --reps 3 > "$out/pmc_k1_$name.log" 2>&1
  python "$root/tools/summarize_pmc.py" "$out/pmc_k1_$name" "$out/r04_k1_pmc_$name.json" "rocprofv3 --pmc $* --kernel-trace -- python tools/bench_embed.py --uniform --vocab 4000000 --reps 3 (2 M ragged lines, uniform ids over a 4 M-row table)" > /dev/null
}
pmc fetch FETCH_SIZE GRBM_GUI_ACTIVE
pmc occ SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
pmc issue SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM
pmc vmem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SALU
pmc tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_REQ_sum
find "$out" -name "*.csv" -size +8M -delete
python - <<'PY'
import json, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04_k1_pmc_*.json")):
    d = json.load(open(f))
    for k, v in d["kernels"].items():
        if "embed_kernel" in k:
            print(f.split("/")[-1], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a.endswith("_avg") or a in ("dispatches", "avg_us_under_pmc")})
PY
step "done"
