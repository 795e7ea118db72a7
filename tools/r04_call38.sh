#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call38
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "rt2d or c1 or resident" ) > $OUT/pytest.txt 2>&1
tail -6 $OUT/pytest.txt | head -3 | cut -c1-300
P() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-110; }
echo "== c1"; P 1 128 1024 2048 --reps 30
echo "== timeline"
timeout 300 python tools/timeline_probe.py c1b 2>&1 | grep -v amdgpu.ids | grep -A12 "C1 batch 128" | tail -13 | cut -c1-200
