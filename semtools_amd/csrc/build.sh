#!/usr/bin/env bash
# Builds semtools_amd/lib/libsemtools_hip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -fno-gpu-rdc)
objs=()
for src in api.cpp scan_kernels.hip embed_kernels.hip gemm_kernels.hip; do
  obj="$out/${src%.*}.o"
  if [[ ! -f "$obj" || "$here/$src" -nt "$obj" || "$here/common.h" -nt "$obj" || "$here/device_utils.h" -nt "$obj" || "$here/../../include/semtools_hip.h" -nt "$obj" ]]; then
    "$HIPCC" "${FLAGS[@]}" -x hip -c "$here/$src" -o "$obj" ${EXTRA_HIPCC_FLAGS:-}
  fi
  objs+=("$obj")
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out/libsemtools_hip.so"
echo "built $out/libsemtools_hip.so"
