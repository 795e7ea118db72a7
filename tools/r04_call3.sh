#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call3
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 600 python -m pytest tests/test_gpu_r2_paths.py -m gpu -q -k rt2d -p no:cacheprovider ) > $OUT/pytest_rt2d.txt 2>&1
tail -12 $OUT/pytest_rt2d.txt
timeout 300 python tools/c1_probe.py 1 8 128 1024 2>&1 | grep -v amdgpu.ids | tee $OUT/c1_probe.txt
timeout 300 python tools/c1_probe.py 128 --grid 400 2>&1 | grep -v amdgpu.ids | tee -a $OUT/c1_probe.txt
timeout 300 python tools/c1_probe.py 128 --dirty 2>&1 | grep -v amdgpu.ids | tee -a $OUT/c1_probe.txt
CMX_SO_PATH=$REPO/cartographer_amd/lib/base_r03/libcartographer_mi355x.so timeout 300 python tools/c1_probe.py 1 128 1024 2>&1 | grep -v amdgpu.ids | sed 's/^/BASE /' | tee -a $OUT/c1_probe.txt
PROFILE_TIMEOUT=200 bash tools/profile_cmd.sh r04_call3/c1b128 "python tools/c1_probe.py 128 --reps 50"
cat $OUT/c1b128_kernel_stats.csv | cut -c1-220 | head -12
PROFILE_TIMEOUT=200 bash tools/profile_cmd.sh r04_call3/c1b1 "python tools/c1_probe.py 1 --reps 100"
cat $OUT/c1b1_kernel_stats.csv | cut -c1-220 | head -12
