#!/bin/bash
# First gpurun call of a round: everything needed to decide where the round's GPU minutes go,
# in one box lease (~6 min).  Logs land in gpurun_out/round_start/.
#   gpurun --timeout 1200 -- 'bash tools/round_start.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/round_start
mkdir -p "$OUT"
cd "$REPO"
export PYTHONUNBUFFERED=1

echo "== gpu tests ==";          ( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5 ) | tee "$OUT/pytest_gpu.txt"
echo "== device vs oracle, random odd-shaped problems ==";  timeout 300 python tools/gpu_fuzz.py 150 1 2>&1 | tail -20 | tee "$OUT/gpu_fuzz.txt"
echo "== bench (default) ==";    timeout 600 python bench.py 2> "$OUT/bench_default.err" | tail -1 | tee "$OUT/bench_default.json" | cut -c1-400
echo "== C2: single stream, 8 Python threads, 8 native threads, C3 share =="
timeout 200 python tools/c2_probe.py 2>&1 | grep "^\[" | tee "$OUT/c2_probe.txt"
echo "== C2 front end: block timeline with wavefront 0's first chunk step by step (DESIGN 5.1) =="
timeout 200 python tools/timeline_probe.py c2 2>&1 | grep -A18 "cmx timeline\] PrepScore" | tail -19 | cut -c1-160 | tee "$OUT/c2_timeline.txt"
echo "== C1: calls of 1 ... 4096 matches; in a process that has served 8 concurrent callers =="
timeout 300 python tools/c1_probe.py 1 128 256 512 1024 2048 4096 --reps 25 2>&1 | grep "^C1" | cut -c1-110 | tee "$OUT/c1_probe.txt"
timeout 300 python tools/probes/c1_after_threads.py 512 1024 2>&1 | grep "^C1" | tee -a "$OUT/c1_probe.txt"
echo "== C4 pass by pass =="
CMX_NO_REPORT=1 timeout 300 python - <<'PY' 2>&1 | grep "cmx trace\|rep 1" | tail -3 | cut -c1-420 | tee "$OUT/c4_trace.txt"
import sys, os, runpy
sys.path.insert(0, os.getcwd())
from cartographer_amd import _lib
_lib.debug_set(trace=1, timing=1)
sys.argv = ["time_rt3d.py", "1"]
runpy.run_path("tools/time_rt3d.py", run_name="__main__")
PY
echo "== C5 single pair / 32-pair share =="
timeout 300 python tools/c5_probe.py 2>&1 | grep "^\[" | tee "$OUT/c5_probe.txt"
echo "== families the bench does not time (histogram, voxel filters, inserters, Ceres) =="
timeout 200 python tools/family_probe.py 2>&1 | grep "us / call" | tee "$OUT/family_probe.txt"
echo "== gather ceilings =="
[ -x tools/bin/row_gather_ceiling ] && timeout 60 ./tools/bin/row_gather_ceiling | tee "$OUT/row_gather_ceiling.txt"
