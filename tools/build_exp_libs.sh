#!/usr/bin/env bash
# Ablation / trace builds of the library for tools/exp_k3.sh and tools/exp_k3_trace.sh: gemm_rowreg.hip compiled with
# -DSMT_RR_EXP=<bits> (gemm_rowreg_kernel: bit 0 no epilogue, 1 no query-tile staging, 2 no ring wait + barrier, 3 row phase only
# in the first step, 8 (256) wave-timeline stamps) and linked with the other objects of the normal build into
# tools/exp_libs/libsemtools_hip_exp<bits>.so.  Usage: bash tools/build_exp_libs.sh 1 7 8 15 256   (after semtools_amd/csrc/build.sh)
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
csrc="$root/semtools_amd/csrc"; lib="$root/semtools_amd/lib"; out="$root/tools/exp_libs"
mkdir -p "$out"
objs=$(ls "$lib"/*.o | grep -v gemm_rowreg.o)
for e in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -x hip -DSMT_RR_EXP="$e" -c "$csrc/gemm_rowreg.hip" -o "/tmp/gemm_exp$e.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $objs "/tmp/gemm_exp$e.o" -ldl -o "$out/libsemtools_hip_exp$e.so"
  echo "built $out/libsemtools_hip_exp$e.so"
done
