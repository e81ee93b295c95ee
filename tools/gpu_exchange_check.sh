#!/usr/bin/env bash
# 1-GPU check of bench.py's N>1 code path: one rank under torch.distributed.run with the exchange forced
# (RCCL all-gather of the packed lists + device merge), pipelined one step deep and not pipelined.
tag="${1:-r01}"
out="$(pwd)/gpurun_out"
mkdir -p "$out"
for pipe in 1 0; do
SEMTOOLS_BENCH_PIPELINE=$pipe SEMTOOLS_BENCH_FORCE_EXCHANGE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 2953$pipe bench.py --gpus 1 --steps 200 --warmup 20 \
    --no-cpu-baseline --no-secondary --no-ivfpq > "$out/bench_exchange_${tag}_pipe$pipe.json" 2> "$out/bench_exchange_${tag}_pipe$pipe.err"
python - "$out/bench_exchange_${tag}_pipe$pipe.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("exchange", sys.argv[1][-10:-5], "ms_per_step", j["ms_per_step"], "host_issue", j["host_issue_ms_per_step"], "scan_us", j["roofline"]["avg_kernel_us"], j["checks"])
PY
done
