#!/bin/bash
# fused front end: cells from the two-FMA f32 estimate
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call49
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 600 python -m pytest tests/test_gpu_2d.py tests/test_gpu_r2_paths.py tests/test_gpu_zz_new.py tests/test_dropin.py -m gpu -q -p no:cacheprovider -x ) > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt | head -2 | cut -c1-300
P2() { timeout 200 python tools/c2_probe.py "$@" 2>&1 | grep "^\[" | grep -v native | cut -c1-200; }
P2 --set timing=1
P2 --no-c3
CMX_SO_PATH=$REPO/cartographer_amd/lib/base_r04/libcartographer_mi355x.so P2 --no-c3 | sed 's/^/BASE /'
P2 --no-c3
