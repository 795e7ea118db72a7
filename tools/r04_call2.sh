#!/bin/bash
# Round 4, RT-2D tile path: parity first, then A/B against the round-3 library (row-pair kernel).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call2
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 600 python -m pytest tests/test_gpu_r2_paths.py -m gpu -q -x -k rt2d -p no:cacheprovider ) > $OUT/pytest_rt2d.txt 2>&1
tail -15 $OUT/pytest_rt2d.txt
( time timeout 600 python -m pytest tests/test_gpu_zz_new.py tests/test_gpu_2d.py tests/test_gpu_grid.py tests/test_dropin.py -m gpu -q -k "rt2d or real_time or grid" -p no:cacheprovider ) > $OUT/pytest_rt2d_more.txt 2>&1
tail -8 $OUT/pytest_rt2d_more.txt
bash tools/ab_r04.sh r04_call2 "python tools/time_configs.py c1 c1b"
