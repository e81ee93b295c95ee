#!/usr/bin/env bash
# K1 counters on the round-6 binary (the round-4 set, re-collected: the kernel got its equal-work runs in round 5): uniform ids over a
# 4 M-row table, 2 M ragged lines.  Separate rocprofv3 --pmc passes, --kernel-trace only beside them.
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pmc() {  # name, counters...
  local name="$1"; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc_k1_$name" -o k1 -- python "$root/tools/bench_embed.py" --uniform --vocab 4000000 --reps 3 > "$out/pmc_k1_$name.log" 2>&1
  python "$root/tools/summarize_pmc.py" "$out/pmc_k1_$name" "$out/r06_k1_pmc_$name.json" "rocprofv3 --pmc $* --kernel-trace -- python tools/bench_embed.py --uniform --vocab 4000000 --reps 3 (2 M ragged lines, uniform ids over a 4 M-row table)" > /dev/null
}
pmc fetch FETCH_SIZE GRBM_GUI_ACTIVE
pmc issue SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM
pmc wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
find "$out" -name "*.csv" -size +8M -delete
python - <<'PY'
import json, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r06_k1_pmc_*.json")):
    d = json.load(open(f))
    for k, v in d["kernels"].items():
        if "embed_kernel" in k:
            print(f.split("/")[-1], k[:48], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a.endswith("_avg") or a in ("dispatches", "avg_us_under_pmc")})
PY
