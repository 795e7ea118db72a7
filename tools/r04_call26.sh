#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call26
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "rt2d or c1" ) > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt | cut -c1-300
P() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-200; }
echo "== default"; P 1 128 1024
for g in 4 2 1; do echo "== groups cap $g"; P 128 1024 --set rt2d_groups=$g; done
echo "== parts 1 / groups 4,2"; P 128 --set rt2d_parts=1 --set rt2d_groups=4; P 128 --set rt2d_parts=1 --set rt2d_groups=2
echo "== 1024 parts 4 groups 2"; P 1024 --set rt2d_parts=4 --set rt2d_groups=2; P 1024 --set rt2d_parts=16 --set rt2d_groups=2
