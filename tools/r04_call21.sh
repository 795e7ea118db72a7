#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call21
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gpu_r2_paths.py tests/test_gpu_zz_new.py tests/test_gpu_grid.py tests/test_gpu_overflow.py -m gpu -q -p no:cacheprovider -k "rt2d" ) > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
BASE=$REPO/cartographer_amd/lib/base_r03/libcartographer_mi355x.so
for R in 1 2; do
  echo "== base"; CMX_SO_PATH=$BASE timeout 300 python tools/c1_probe.py 1 128 1024 2>&1 | grep -v amdgpu.ids
  echo "== new (fused, inline finish)"; timeout 300 python tools/c1_probe.py 1 128 1024 2>&1 | grep -v amdgpu.ids
  echo "== new copy late"; timeout 300 python tools/c1_probe.py 128 1024 --set rt2d_copy_late=1 2>&1 | grep -v amdgpu.ids; echo "== new unfused"; timeout 300 python tools/c1_probe.py 128 --set rt2d_unfused=1 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $OUT/ab.txt
echo "== grid 400, dirty"; timeout 300 python tools/c1_probe.py 128 --grid 400 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt; timeout 300 python tools/c1_probe.py 128 --dirty 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
echo "== timeline b128 (one part)"; timeout 120 python tools/c1_probe.py 128 --reps 2 --set timeline=1 --set rt2d_parts=1 2>&1 | grep -v amdgpu.ids | tail -20 > $OUT/timeline_b128.txt; cat $OUT/timeline_b128.txt
PROFILE_TIMEOUT=200 bash tools/profile_cmd.sh r04_call21/c1b128 "python tools/c1_probe.py 128 --reps 50" > /dev/null
cat $OUT/c1b128_kernel_stats.csv | sed 's/cmx::(anonymous namespace):://g' | awk -F'"' '{print substr($2,1,40), $3}' | head -8
echo "== timeline copy late"; timeout 120 python tools/c1_probe.py 128 --reps 2 --set timeline=1 --set rt2d_parts=1 --set rt2d_copy_late=1 2>&1 | grep -v amdgpu.ids | grep -A9 "Rt2DTileKernel"
