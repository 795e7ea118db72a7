#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
for g in 4 2 1; do
echo "== timeline groups<=$g"
timeout 300 python - $g <<'PY' 2>&1 | grep -v amdgpu.ids | grep -B1 -A10 "Rt2DTileKernel" | tail -12 | cut -c1-160
import sys, os, runpy
sys.path.insert(0, os.getcwd())
from cartographer_amd import _lib
_lib.debug_set(rt2d_groups=int(sys.argv[1]), rt2d_parts=1)
sys.argv = ["timeline_probe.py", "c1b"]
runpy.run_path("tools/timeline_probe.py", run_name="__main__")
PY
done
