#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-260; }
echo "== 128 parts 1"; P 128 --reps 4 --set host_trace=1 --set rt2d_parts=1 | grep -v "tile kernel" | tail -13
echo "== 1024 parts 2"; P 1024 --reps 3 --set host_trace=1 --set rt2d_parts=2 | grep -v "tile kernel" | tail -16
echo "== 1024 parts 1"; P 1024 --reps 3 --set host_trace=1 --set rt2d_parts=1 | grep -v "tile kernel" | tail -9
