#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call25
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "rt2d or c1 or resident or dropin" ) > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt | cut -c1-300
echo "== probe"; timeout 300 python tools/c1_probe.py 1 128 1024 2>&1 | grep -v amdgpu.ids | cut -c1-400
echo "== probe g400"; timeout 300 python tools/c1_probe.py 128 --grid 400 2>&1 | grep -v amdgpu.ids | cut -c1-400
echo "== probe dirty"; timeout 300 python tools/c1_probe.py 128 --dirty 2>&1 | grep -v amdgpu.ids | cut -c1-400
echo "== timeline"; timeout 300 python tools/timeline_probe.py c1b 2>&1 | grep -v amdgpu.ids | grep -A12 "C1 batch 128" | tail -30 | cut -c1-400
