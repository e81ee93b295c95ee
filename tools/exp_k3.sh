#!/usr/bin/env bash
# K3 ablations: each tools/exp_libs/libsemtools_hip_exp<bits>.so is the library with parts of gemm_rowreg_kernel compiled out
# (bit 0: epilogue, 1: query-tile staging, 2: ring wait + barrier, 3: row load/convert after the first step).  Results are
# WRONG by construction; only gemm_ms is read.
cp semtools_amd/lib/libsemtools_hip.so /tmp/orig.so
for e in 0 $@; do
  if [ "$e" = 0 ]; then cp /tmp/orig.so semtools_amd/lib/libsemtools_hip.so; else cp tools/exp_libs/libsemtools_hip_exp$e.so semtools_amd/lib/libsemtools_hip.so; fi
  echo "exp $e"; timeout 200 python tools/bench_small_batch.py --nq 1000 --reps 5 $EXP_ARGS 2>&1 | grep -o '"nq": [0-9]*\|"gemm_ms": [0-9.]*\|"gemm_launches": [0-9]*' | paste - - -
done
cp /tmp/orig.so semtools_amd/lib/libsemtools_hip.so
