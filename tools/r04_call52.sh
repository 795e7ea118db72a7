#!/bin/bash
# (record: fast_2d.hip at the time of this call held the pipelined-gather variant that was removed -- DESIGN.md 5.1)
# fused front end: branch-free accumulation, half-chunk gathers kept in flight across chunks
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call52
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 600 python -m pytest tests/test_gpu_2d.py tests/test_gpu_r2_paths.py tests/test_gpu_zz_new.py -m gpu -q -p no:cacheprovider -x ) > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt | head -2 | cut -c1-300
P2() { timeout 200 python tools/c2_probe.py "$@" 2>&1 | grep "^\[" | grep -v native | cut -c1-200; }
P2
P2 --no-c3
timeout 200 python tools/timeline_probe.py c2 2>&1 | grep -v amdgpu.ids | grep -A9 "cmx timeline\] PrepScore" | tail -10 | cut -c1-200
