#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=gpurun_out/r05_call12; mkdir -p $OUT
export PYTHONUNBUFFERED=1
echo "== gpu tests"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
echo "== C1 on 400 x 400 grids: bounds, tiles"
timeout 300 python tools/c1_probe.py 128 256 --grid 400 --reps 25 --set rt2d_bounds=1 2>&1 | grep "^C1" | cut -c1-200 | tee $OUT/c1_probe.txt
timeout 300 python tools/c1_probe.py 128 256 --grid 400 --reps 25 --set rt2d_no_bounds=1 2>&1 | grep "^C1" | cut -c1-200 | tee -a $OUT/c1_probe.txt
timeout 300 python tools/c1_probe.py 128 --grid 400 --reps 2 --set host_trace=1 2>&1 | grep "bound kernel" | head -2
