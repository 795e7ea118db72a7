#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call34
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "rt2d or c1 or resident" ) > $OUT/pytest.txt 2>&1
tail -6 $OUT/pytest.txt | head -3 | cut -c1-300
P() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-110; }
echo "== default"; P 1 64 128 256 512 1024 2048 --reps 30
echo "== 1024 equal parts 2/4"; P 1024 --reps 30 --set rt2d_parts=2; P 1024 --reps 30 --set rt2d_parts=4
echo "== 128 g400 / dirty"; P 128 --reps 30 --grid 400; P 128 --reps 30 --dirty
echo "== trace 1024"; timeout 300 python tools/c1_probe.py 1024 --reps 2 --set host_trace=1 2>&1 | grep "part\|batch" | tail -10 | cut -c1-200
