#!/bin/bash
# Every device test + the C1 probe (round 4).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_full
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $OUT/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" $OUT/pytest_gpu.txt | tail -3
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.txt | head -20
timeout 300 python tools/c1_probe.py 1 128 1024 2>&1 | grep -v amdgpu.ids | tee $OUT/c1_probe.txt
timeout 300 python tools/c1_probe.py 128 --grid 400 2>&1 | grep -v amdgpu.ids | tee -a $OUT/c1_probe.txt
CMX_SO_PATH=$REPO/cartographer_amd/lib/base_r03/libcartographer_mi355x.so timeout 300 python tools/c1_probe.py 128 --grid 400 2>&1 | grep -v amdgpu.ids | sed 's/^/BASE /' | tee -a $OUT/c1_probe.txt
