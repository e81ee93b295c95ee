#!/usr/bin/env bash
# round 5 closing evidence (ONE run): the default bench line + detail; per-leg rocprofv3 kernel traces reduced to one row per
# (kernel, launch shape) by tools/kernel_shapes.py; the group issue figures per shard count; the CLI end to end with and without
# write-ahead; the sharded IVF figures with every call's time.
tag="${1:-r05}"
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
timeout 900 python bench.py --detail-out "$out/${tag}_bench_detail.json" > "$out/${tag}_bench_line.json" 2> "$out/${tag}_bench.err"; echo "bench rc=$? line bytes=$(tail -1 "$out/${tag}_bench_line.json" | wc -c)"
cd /tmp && export TMPDIR=/tmp
all_off="--no-cpu-baseline --no-c4 --no-secondary --no-embed --no-ivfpq --no-workspace --no-ingest --no-group-issue"
for leg in c2 c4 c3 embed workspace ivfpq; do
  case $leg in
    c2) flags="$all_off --steps 1000 --warmup 100";;
    c4) flags="${all_off/--no-c4/} --steps 20 --warmup 5";;
    c3) flags="${all_off/--no-secondary/} --steps 20 --warmup 5";;
    embed) flags="${all_off/--no-embed/} --steps 20 --warmup 5";;
    workspace) flags="${all_off/--no-workspace/} --steps 20 --warmup 5";;
    ivfpq) flags="${all_off/--no-ivfpq/} --c5-full-rows 0 --steps 20 --warmup 5";;
  esac
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}_$leg" -o bench -- python "$root/bench.py" $flags --detail-out "$out/${tag}_bench_detail_prof_$leg.json" > "$out/prof_${tag}_$leg.log" 2>&1
  trace=$(find "$out/prof_${tag}_$leg" -name "*kernel_trace.csv" | head -1)
  stats=$(find "$out/prof_${tag}_$leg" -name "*kernel_stats.csv" | head -1)
  [ -n "$stats" ] && cp "$stats" "$out/${tag}_bench_${leg}_kernel_stats.csv"
  [ -n "$trace" ] && python "$root/tools/kernel_shapes.py" "$trace" > "$out/${tag}_bench_${leg}_kernel_shapes.csv" && head -4 "$out/${tag}_bench_${leg}_kernel_shapes.csv" | cut -c1-160
done
cd "$root"
find "$out" -name "*kernel_trace.csv" -size +1M -delete
timeout 600 python - <<'PY' > "$out/${tag}_group_issue.json" 2> "$out/${tag}_group_issue.err"
import json, sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
import semtools_amd as smt
res = {}
for n in (1, 2, 4, 8):
    r = bench.bench_group_issue(smt, torch.device("cuda", 0), n_shards=n)
    res[str(n)] = {k: r[k] for k in ("host_issue_us_per_search", "end_to_end_us_per_search", "every_rank_wants_the_answer_us", "copy_transport_us",
                                     "copy_transport_every_rank_us", "one_thread_issues_every_shard_us", "checks")}
print(json.dumps(res, indent=1))
PY
echo "group issue rc=$?"
for wa in 1 0; do SEMTOOLS_WRITE_AHEAD=$wa timeout 600 python tools/bench_cli.py > "$out/${tag}_cli_wa$wa.json" 2> "$out/${tag}_cli_wa$wa.err"; echo "cli wa=$wa rc=$?"; done
timeout 900 python tools/bench_sharded_ivf.py > "$out/${tag}_sharded_ivf.json" 2> "$out/${tag}_sharded_ivf.err"; echo "sharded ivf rc=$?"
tail -1 "$out/${tag}_bench_line.json" | cut -c1-1500
