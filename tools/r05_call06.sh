#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=gpurun_out/r05_call06; mkdir -p $OUT
export PYTHONUNBUFFERED=1
echo "== rt2d tests"; timeout 900 python -m pytest tests/test_gpu_r2_paths.py tests/test_gpu_2d.py tests/test_gpu_zz_new.py -m gpu -q -p no:cacheprovider -k "rt2d" 2>&1 | tail -3
echo "== occupancy"
timeout 300 python tools/c1_probe.py 128 --reps 2 --set host_trace=1 2>&1 | grep "bound kernel" | head -3 | tee $OUT/occupancy.txt
echo "== timeline"
timeout 300 python tools/timeline_probe.py c1 c1b 2>&1 | grep -v "^$" | tail -34 | cut -c1-200 | tee $OUT/timeline.txt
echo "== C1 call sizes: bounds, then tiles"
timeout 300 python tools/c1_probe.py 1 16 128 1024 --reps 25 2>&1 | grep "^C1" | cut -c1-200 | tee $OUT/c1_probe.txt
timeout 300 python tools/c1_probe.py 1 16 128 1024 --reps 25 --set rt2d_no_bounds=1 2>&1 | grep "^C1" | cut -c1-200 | tee -a $OUT/c1_probe.txt
