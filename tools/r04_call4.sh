#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call4
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 600 python -m pytest tests/test_gpu_r2_paths.py -m gpu -q -k rt2d -p no:cacheprovider ) > $OUT/pytest_rt2d.txt 2>&1
tail -6 $OUT/pytest_rt2d.txt
timeout 300 python tools/c1_probe.py 1 128 1024 2>&1 | grep -v amdgpu.ids | tee $OUT/c1_probe.txt
echo "== timeline b1"; timeout 120 python tools/c1_probe.py 1 --reps 2 --set timeline=1 2>&1 | grep -v amdgpu.ids | tail -12 | tee $OUT/timeline_b1.txt
echo "== timeline b64 (one part)"; timeout 120 python tools/c1_probe.py 64 --reps 2 --set timeline=1 --set rt2d_parts=1 2>&1 | grep -v amdgpu.ids | tail -12 | tee $OUT/timeline_b64.txt
echo "== timeline b512 (one part)"; timeout 120 python tools/c1_probe.py 512 --reps 2 --set timeline=1 --set rt2d_parts=1 2>&1 | grep -v amdgpu.ids | tail -12 | tee $OUT/timeline_b512.txt
