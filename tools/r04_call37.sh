#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call37
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "rt2d or rt3d or c1 or resident" ) > $OUT/pytest.txt 2>&1
tail -6 $OUT/pytest.txt | head -3 | cut -c1-300
P() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-110; }
echo "== c1"; P 1 128 1024 --reps 30
echo "== C4"; CMX_NO_REPORT=1 timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | grep "rep 1\|trace" | tail -3 | cut -c1-420
import sys, os, runpy
sys.path.insert(0, os.getcwd())
from cartographer_amd import _lib
_lib.debug_set(trace=1)
sys.argv = ["time_rt3d.py", "1"]
runpy.run_path("tools/time_rt3d.py", run_name="__main__")
PY
