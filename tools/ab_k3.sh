#!/usr/bin/env bash
# Same-box A/B of two builds of the library on the batched path: semtools_amd/lib/ab/libA.so / libB.so copied over
# the product library in turn, tools/k3_call.py times 1000 x 10 M device-resident calls over the image and over the f32 rows, three
# rounds each; then the row-phase ablation (tools/exp_libs/libsemtools_hip_exp8.so: rows loaded + converted in the first step only --
# WRONG answers, the upper bound of what hiding the row phase behind other waves' MFMAs could win).
set -u
out=gpurun_out; mkdir -p $out
LIB=semtools_amd/lib/libsemtools_hip.so
cp $LIB /tmp/lib_orig.so
for round in 1 2 3; do
  for v in A B; do
    cp semtools_amd/lib/ab/lib$v.so $LIB
    echo "$v $round image   $(timeout 200 python tools/k3_call.py --nq 1000 --reps 10 2>/dev/null | tail -1)"
    echo "$v $round f32rows $(timeout 200 python tools/k3_call.py --nq 1000 --reps 10 --no-image 2>/dev/null | tail -1)"
  done
done
if [ -f tools/exp_libs/libsemtools_hip_exp8.so ]; then
  for round in 1 2; do
    cp /tmp/lib_orig.so $LIB
    echo "full     $round f32rows $(timeout 200 python tools/k3_call.py --nq 1000 --reps 10 --no-image 2>/dev/null | tail -1)"
    cp tools/exp_libs/libsemtools_hip_exp8.so $LIB
    echo "no-rowph $round f32rows $(timeout 200 python tools/k3_call.py --nq 1000 --reps 10 --no-image 2>/dev/null | tail -1)"
  done
fi
cp /tmp/lib_orig.so $LIB
