#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call24
mkdir -p $OUT
export PYTHONUNBUFFERED=1
run() { timeout 300 python - "$@" <<'PY' 2>&1 | grep -v amdgpu.ids | cut -c1-900
import sys, os, runpy
sys.path.insert(0, os.getcwd())
from cartographer_amd import _lib
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:]}
if kw: _lib.debug_set(**kw)
os.environ["CMX_NO_REPORT"] = "1"
sys.argv = ["time_rt3d.py", "1"]
runpy.run_path("tools/time_rt3d.py", run_name="__main__")
PY
}
echo "== rotblocks trace"; run trace=1 | grep -v "^bulk 1 rep 0" | tail -40
echo "== dense trace"; run trace=1 rt3d_no_rotblocks=1 | tail -25
for p in 900 950 990; do echo "== permille $p"; run rt3d_rotblock_permille=$p | tail -1 | cut -c1-260; done
PROFILE_TIMEOUT=200 bash tools/profile_cmd.sh r04_call24/c4 "python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-other" > /dev/null
cat $OUT/c4_kernel_stats.csv | sed 's/cmx::(anonymous namespace):://g' | awk -F'"' '{print substr($2,1,60), $3}' | head -16
