#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call16
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gpu_r2_paths.py tests/test_gpu_zz_new.py tests/test_gpu_grid.py tests/test_dropin.py -m gpu -q -p no:cacheprovider -k "rt2d or intensity or sharded" ) > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
timeout 300 python tools/c1_probe.py 1 128 1024 2>&1 | grep -v amdgpu.ids | tee $OUT/c1_probe.txt
timeout 300 python tools/c1_probe.py 128 --reps 3 --set host_trace=1 2>&1 | grep -v amdgpu.ids | grep "tile kernel" | tail -1
echo "== grid 400, dirty"; timeout 300 python tools/c1_probe.py 128 --grid 400 2>&1 | grep -v amdgpu.ids | tee -a $OUT/c1_probe.txt; timeout 300 python tools/c1_probe.py 128 --dirty 2>&1 | grep -v amdgpu.ids | tee -a $OUT/c1_probe.txt
echo "== timeline b128 (one part)"; timeout 120 python tools/c1_probe.py 128 --reps 2 --set timeline=1 --set rt2d_parts=1 2>&1 | grep -v amdgpu.ids | tail -18 > $OUT/timeline_b128.txt; cat $OUT/timeline_b128.txt
PROFILE_TIMEOUT=200 bash tools/profile_cmd.sh r04_call16/c1b128 "python tools/c1_probe.py 128 --reps 50" > /dev/null
cat $OUT/c1b128_kernel_stats.csv | sed 's/cmx::(anonymous namespace):://g' | awk -F'"' '{print substr($2,1,40), $3}' | head -8
