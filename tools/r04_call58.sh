#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "rt2d or c1 or resident" ) 2>&1 | tail -2
P1() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | grep "^C1" | cut -c17-110; }
echo "== default policy"; P1 128 192 256 384 512 768 1024 1536 2048 4096 --reps 25
echo "== 400x400"; P1 256 1024 --reps 25 --grid 400
echo "== dirty"; P1 256 1024 --reps 25 --dirty
