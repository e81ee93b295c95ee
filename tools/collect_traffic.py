"""Fold a tools/summarize_pmc.py summary (rocprofv3 --pmc FETCH_SIZE --kernel-trace of a bench.py run) into
profiles/traffic.json: for every bench leg the corrected HBM bytes per launch of its dominant kernel, with the source.

    python tools/collect_traffic.py <summary.json> <tag, e.g. profiles/r03_pmc_fetch.json> "<command that was profiled>"

Leg -> kernel name fragment, and which statistic of the dispatches is the leg's launch: the LARGEST launch of the kernel
(the c3 main level, the c4-size scan) or the average (c2: every launch scans the same 1 M-row shard)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEGS = {
    # leg: (kernel fragment, statistic, rows key)
    "c2": ("scan_topk_kernel<1, 4, true, false,", "min"),   # (the same kernel also runs c4's 100 M-row launches: min = the 1 M-row shard)
    "c4": ("scan_topk_kernel<1, 4, true, false,", "max"),
    "c3": ("gemm_rowreg_kernel<", "max"),   # main level of the batch (15/16 of the corpus)
    "embed_zipf_ids_500k_table": ("embed_kernel", "min"),
    "embed_uniform_ids_4M_table": ("embed_kernel", "max"),
    "ivf_adc_lpca": ("ivf_adc_kernel", "avg"),   # (one kernel template for both codings: profile them in separate runs)
    "ivf_adc_pq": ("ivf_adc_kernel", "avg"),
}


def main():
    summary, tag, command = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {}
    kernels = json.load(open(summary))["kernels"]
    path = os.path.join(ROOT, "profiles", "traffic.json")
    doc = json.load(open(path))
    for leg, (frag, stat) in LEGS.items():
        if leg not in rows:
            continue
        hit = [v for k, v in kernels.items() if frag in k and "FETCH_SIZE_avg" in v]
        if not hit:
            continue
        v = hit[0]
        kb = {"avg": v["FETCH_SIZE_avg"], "max": v["FETCH_SIZE_max"], "min": v.get("FETCH_SIZE_min", v["FETCH_SIZE_avg"])}[stat]
        doc["legs"][leg] = {"rows": rows[leg]["rows"], "kernel": [k for k in kernels if frag in k][0],
                            "hbm_bytes_per_launch": 2048.0 * kb, "algorithmic_bytes_per_launch": rows[leg].get("algorithmic_bytes"),
                            "ratio": (2048.0 * kb / rows[leg]["algorithmic_bytes"]) if rows[leg].get("algorithmic_bytes") else None,
                            "statistic": f"{stat} over {v['dispatches']} dispatches",
                            "counter": "rocprofv3 --pmc FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B; MI355X_MICROARCH.md section HBM)",
                            "source": f"{tag} <- {command}"}
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(doc["legs"], indent=1)[:4000])


if __name__ == "__main__":
    main()
