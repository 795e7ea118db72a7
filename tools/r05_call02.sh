#!/bin/bash
# Round 5, call 2: the block-bound level of RT-2D -- its invariant, parity on every path, timing.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=gpurun_out/r05_call02; mkdir -p $OUT
export PYTHONUNBUFFERED=1
echo "== block bounds dominate (verify mode)"
timeout 600 python -m pytest tests/test_gpu_r2_paths.py -m gpu -q -p no:cacheprovider -x -k "block_bounds" 2>&1 | tail -15 | tee $OUT/verify.txt
echo "== rt2d tests"
timeout 900 python -m pytest tests/test_gpu_r2_paths.py tests/test_gpu_2d.py tests/test_gpu_zz_new.py -m gpu -q -p no:cacheprovider -k "rt2d" 2>&1 | tail -15 | tee $OUT/rt2d_tests.txt
echo "== C1 call sizes: bounds, then tiles"
timeout 300 python tools/c1_probe.py 1 128 1024 --reps 25 2>&1 | grep "^C1" | cut -c1-200 | tee $OUT/c1_probe.txt
timeout 300 python tools/c1_probe.py 1 128 1024 --reps 25 --set rt2d_no_bounds=1 2>&1 | grep "^C1" | cut -c1-200 | tee -a $OUT/c1_probe.txt
