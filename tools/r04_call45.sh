#!/bin/bash
# (record of an experiment: the debug switch it sets was removed together with the variant that lost -- DESIGN.md 5.3)
# chained tile kernels of a batch's parts
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P1() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | grep "^C1" | cut -c1-120; }
echo "== chain"; P1 256 512 1024 2048 --reps 30
echo "== no chain"; P1 256 512 1024 2048 --reps 30 --set rt2d_no_chain=1
echo "== chain"; P1 1024 --reps 30
echo "== chain, 8 queues"; GPU_MAX_HW_QUEUES=8 P1 1024 2048 --reps 30
echo "== trace"; timeout 300 python tools/c1_probe.py 1024 --reps 2 --set host_trace=1 2>&1 | grep "collected\|batch(" | tail -5 | cut -c1-200
( timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "rt2d or c1 or resident" ) 2>&1 | tail -2
