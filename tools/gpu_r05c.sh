#!/usr/bin/env bash
# round 5: the c5 leg after the ADC kernel was split by coding kind (occupancy), PQ with 4- and 8-wave blocks
out="$(pwd)/gpurun_out"; mkdir -p "$out"
for w in 4; do
SEMTOOLS_IVF_PQ_WAVES=$w timeout 600 python bench.py --steps 50 --warmup 10 --no-secondary --no-embed --no-workspace --no-ingest --no-group-issue --no-c4 --no-cpu-baseline --c5-full-rows 0 \
   --detail-out "$out/r05c_ivf_detail_w$w.json" > "$out/r05c_ivf_line_w$w.json" 2> "$out/r05c_ivf_w$w.err"
python - "$out/r05c_ivf_detail_w$w.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["ivfpq"]
for name, leg in (("lpca", d), ("pq", d["global_pq_m32"])):
    r = leg["roofline"]
    print(name, "recall", leg["recall_at_k_vs_exact"], "q/s", round(leg["queries_per_s"]), "adc_ms", round(r["adc_ms_per_batch"], 4), "frac", round(r["frac"], 4), "probe_ms", round(r["probe_ms_per_batch"], 4))
PY
done
