#!/bin/bash
# direct results (no copy kernel behind the last launch), one upload in fast-3D, no_timing A/B
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call40
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 600 python -m pytest tests/test_gpu_2d.py tests/test_gpu_r2_paths.py tests/test_gpu_zz_new.py tests/test_gpu_3d.py -m gpu -q -p no:cacheprovider -x ) > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt | head -2 | cut -c1-300
P2() { timeout 200 python tools/c2_probe.py "$@" 2>&1 | grep "^\[" | cut -c1-200; }
P5() { timeout 200 python tools/c5_probe.py "$@" 2>&1 | grep "^\[" | cut -c1-200; }
P2
P2 --set no_direct_results=1 --no-c3
P2 --set no_timing=1 --no-c3
P2 --set no_timing=1 --set no_direct_results=1 --no-c3
P2 --no-c3
P5
P5 --set no_direct_results=1
P5 --set no_timing=1
P5
