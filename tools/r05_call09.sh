#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=gpurun_out/r05_call09; mkdir -p $OUT
export PYTHONUNBUFFERED=1
echo "== rt2d tests"; timeout 900 python -m pytest tests/test_gpu_r2_paths.py tests/test_gpu_2d.py tests/test_gpu_zz_new.py -m gpu -q -p no:cacheprovider -k "rt2d" 2>&1 | tail -3
for ARGS in "1" "128" "512 rt2d_parts=1"; do
  echo "== timeline $ARGS"
  timeout 120 python tools/probes/c1_timeline.py $ARGS 2>&1 | grep -A12 "Rt2DBoundKernel" | grep -v Finish | head -12 | cut -c1-150
done 2>&1 | tee $OUT/timeline.txt
echo "== C1 call sizes: bounds, then tiles"
timeout 300 python tools/c1_probe.py 1 16 128 1024 --reps 25 2>&1 | grep "^C1" | cut -c1-200 | tee $OUT/c1_probe.txt
timeout 300 python tools/c1_probe.py 1 16 128 1024 --reps 25 --set rt2d_no_bounds=1 2>&1 | grep "^C1" | cut -c1-200 | tee -a $OUT/c1_probe.txt
