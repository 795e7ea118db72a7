#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-110; }
for V in "" var_vgpr "" var_vgpr; do
  if [ -n "$V" ]; then export CMX_SO_PATH=$REPO/cartographer_amd/lib/$V/libcartographer_mi355x.so; else unset CMX_SO_PATH; fi
  echo "== ${V:-product}"; P 128 1024 --reps 40
done
echo "== timeline vgpr"
timeout 300 python tools/timeline_probe.py c1b 2>&1 | grep -v amdgpu.ids | grep -A12 "C1 batch 128" | tail -13 | cut -c1-200
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "rt2d" 2>&1 | tail -2
