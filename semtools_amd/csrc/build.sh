#!/usr/bin/env bash
# Builds semtools_amd/lib/libsemtools_hip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -fno-gpu-rdc)
objs=()
pids=()
jobs_max="${BUILD_JOBS:-$(nproc)}"
newest_hdr="$(ls -t "$here"/*.h "$here"/host/*.h "$here"/../../include/*.h | head -1)"
for src in api.cpp search.cpp corpus_io.cpp group.cpp sharded.cpp scan_kernels.hip embed_kernels.hip gemm_topk.hip gemm_rowreg.hip gemm_ldsrow.hip gemm_level.hip largek.hip threshold.hip domain.hip ivfpq_build.hip ivfpq_search.hip ivfpq_io.hip host/host.cpp host/store.cpp host/output.cpp host/hf_tokenizer.cpp host/host_capi.cpp; do
  base="$(basename "${src%.*}")"
  obj="$out/$base.o"
  if [[ ! -f "$obj" || "$here/$src" -nt "$obj" || "$newest_hdr" -nt "$obj" ]]; then
    "$HIPCC" "${FLAGS[@]}" -x hip -c "$here/$src" -o "$obj" ${EXTRA_HIPCC_FLAGS:-} &
    pids+=($!)
    while (( $(jobs -rp | wc -l) >= jobs_max )); do wait -n || { echo "compile failed" >&2; exit 1; }; done
  fi
  objs+=("$obj")
done
for pid in "${pids[@]}"; do wait "$pid" || { echo "compile failed" >&2; exit 1; }; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -pthread "${objs[@]}" -ldl -o "$out/libsemtools_hip.so"
echo "built $out/libsemtools_hip.so"
# CLI replica (host-only C++), finds the library next to it
bin="$here/../bin"
mkdir -p "$bin"
g++ -O2 -std=c++17 -Wall "$here/host/cli.cpp" -o "$bin/semtools" -L"$out" -lsemtools_hip -Wl,-rpath,'$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib
echo "built $bin/semtools"
