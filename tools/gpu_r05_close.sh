#!/usr/bin/env bash
# The last refresh of round 5 (after the sweeps changed K1's runs, the level plan of mid-sized corpora, the routing borders and the
# nomination mode of 129..255 queries): the default bench line + detail, the embed leg's kernel shapes (embed_runs_kernel is new),
# the group issue figures.  Everything else under profiles/r05_* is from tools/gpu_r05_final.sh and unaffected (c2 / c3 / c4 / IVF
# kernels and their launch shapes did not change).
tag="${1:-r05}"
root="$(pwd)"; out="$root/gpurun_out"; mkdir -p "$out"
timeout 900 python bench.py --detail-out "$out/${tag}_bench_detail.json" > "$out/${tag}_bench_line.json" 2> "$out/${tag}_bench.err"; echo "bench rc=$? line bytes=$(tail -1 "$out/${tag}_bench_line.json" | wc -c)"
cd /tmp && export TMPDIR=/tmp
all_off="--no-cpu-baseline --no-c4 --no-secondary --no-embed --no-ivfpq --no-workspace --no-ingest --no-group-issue"
for leg in embed; do
  flags="${all_off/--no-embed/} --steps 20 --warmup 5"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}_$leg" -o bench -- python "$root/bench.py" $flags --detail-out "$out/${tag}_bench_detail_prof_$leg.json" > "$out/prof_${tag}_$leg.log" 2>&1
  trace=$(find "$out/prof_${tag}_$leg" -name "*kernel_trace.csv" | head -1)
  [ -n "$trace" ] && python "$root/tools/kernel_shapes.py" "$trace" > "$out/${tag}_bench_${leg}_kernel_shapes.csv" && head -6 "$out/${tag}_bench_${leg}_kernel_shapes.csv" | cut -c1-160
done
cd "$root"
find "$out" -name "*kernel_trace.csv" -size +1M -delete
timeout 600 python - <<'PY' > "$out/${tag}_group_issue_final.json" 2> "$out/${tag}_group_issue.err"
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
tail -1 "$out/${tag}_bench_line.json" | cut -c1-1800
