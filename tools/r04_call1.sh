#!/bin/bash
# Round 4, first GPU call: the gathers-in-flight rewrite of the tree-search kernels
# (fast_2d: ExpandWaveKernel / ScoreChildren; fast_3d: FamilySums3D / ChildSums3D).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call1
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 600 python -m pytest tests/test_gpu_2d.py tests/test_gpu_r2_paths.py tests/test_gpu_zz_new.py tests/test_gpu_3d.py -m gpu -q -x -p no:cacheprovider ) > $OUT/pytest_fast.txt 2>&1
tail -5 $OUT/pytest_fast.txt
bash tools/ab_r04.sh r04_call1 "python tools/c2_probe.py" "python tools/time_configs.py c5" "python tools/time_c5.py 32 | head -3"
( time timeout 300 python tools/stress_fast3d.py 80 ) > $OUT/stress_fast3d.txt 2>&1
tail -4 $OUT/stress_fast3d.txt
