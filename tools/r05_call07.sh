#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=gpurun_out/r05_call07; mkdir -p $OUT
export PYTHONUNBUFFERED=1
for V in "" bexp1 bexp2 bexp3; do
  SO=cartographer_amd/lib/libcartographer_mi355x.so; [ -n "$V" ] && SO=cartographer_amd/lib/var_$V/libcartographer_mi355x.so
  for ARGS in "256 rt2d_parts=1" "512 rt2d_parts=1"; do
    echo "== variant '${V:-product}' $ARGS"
    CMX_SO_PATH=$PWD/$SO timeout 120 python tools/probes/c1_timeline.py $ARGS 2>&1 | grep -A16 "Rt2DBoundKernel" | grep -v Finish | head -16 | cut -c1-150
  done
done 2>&1 | tee $OUT/timeline_variants.txt
