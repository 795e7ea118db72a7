#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call30
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "rt2d or c1" ) > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt | cut -c1-300
P() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-150; }
for V in "" var_t128 var_j4 "" var_t128 var_j4; do
  if [ -n "$V" ]; then export CMX_SO_PATH=$REPO/cartographer_amd/lib/$V/libcartographer_mi355x.so; else unset CMX_SO_PATH; fi
  echo "== ${V:-product}"; P 128 1024 --reps 40
done
unset CMX_SO_PATH
echo "== timeline product"
timeout 300 python tools/timeline_probe.py c1b 2>&1 | grep -v amdgpu.ids | grep -A12 "C1 batch 512" | tail -13 | cut -c1-200
