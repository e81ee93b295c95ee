#!/usr/bin/env bash
# K3 wave timeline: runs tools/trace_k3.py with the trace build of the library (tools/exp_libs/libsemtools_hip_exp256.so).
cp semtools_amd/lib/libsemtools_hip.so /tmp/orig.so
cp tools/exp_libs/libsemtools_hip_exp256.so semtools_amd/lib/libsemtools_hip.so
timeout 200 python tools/trace_k3.py "$@" > /tmp/trace.out 2> /tmp/trace.err; tail -3 /tmp/trace.err; tail -1 /tmp/trace.out > gpurun_out/r03_k3_wave_timeline.json
cp /tmp/orig.so semtools_amd/lib/libsemtools_hip.so
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_k3_wave_timeline.json"))
print("shader clock over 8 steps of the traced block: %s GHz" % d.get("shader_clock_GHz_over_8_steps"))
for w in ("0", "4"):
    v = d["waves"][w]
    print("wave", w, "total", v["total_ticks"])
    for k, t in v["transitions"].items():
        print("   %-40s n %3d mean %8.1f max %6d sum %7d" % (k, t["n"], t["mean"], t["max"], t["sum"]))
print("barrier arrivals relative to the last arrival, per wave (rows) and barrier (columns):")
arr = d["barrier_arrive"]
nb = min(len(v) for v in arr.values())
for w in sorted(arr):
    print(w, [arr[w][k] - max(arr[x][k] for x in arr) for k in range(nb)])
print("last arrival -> release of wave 0:", [d["barrier_release"]["0"][k] - max(arr[x][k] for x in arr) for k in range(nb)])
print("barrier period:", [d["barrier_release"]["0"][k + 1] - d["barrier_release"]["0"][k] for k in range(nb - 1)])
print("first product start per wave:", {w: v[0] for w, v in d["product_start"].items()})
PY
