#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call23
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 1200 python -m pytest tests/test_gpu_3d.py tests/test_gpu_zz_new.py -m gpu -q -p no:cacheprovider -x -k "rt3d" ) > $OUT/pytest.txt 2>&1
tail -25 $OUT/pytest.txt | cut -c1-300
echo "== C4 timing: rotblocks (default)"; timeout 300 python tools/time_rt3d.py 1 2>&1 | grep -v amdgpu.ids | cut -c1-700
echo "== C4 timing: dense"; timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | cut -c1-700
import sys, os, runpy
sys.path.insert(0, os.getcwd())
from cartographer_amd import _lib
_lib.debug_set(rt3d_no_rotblocks=1)
sys.argv = ["time_rt3d.py", "1"]
runpy.run_path("tools/time_rt3d.py", run_name="__main__")
PY
echo "== no report"; CMX_NO_REPORT=1 timeout 300 python tools/time_rt3d.py 1 2>&1 | grep -v amdgpu.ids | cut -c1-300
