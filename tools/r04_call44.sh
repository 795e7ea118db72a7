#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
echo "== trace 1024"; timeout 300 python tools/c1_probe.py 1024 --reps 3 --set host_trace=1 2>&1 | grep -v amdgpu.ids | tail -40 | cut -c1-260
echo "== trace 128"; timeout 300 python tools/c1_probe.py 128 --reps 3 --set host_trace=1 2>&1 | grep -v amdgpu.ids | tail -14 | cut -c1-260
echo "== 1 match"; timeout 300 python tools/c1_probe.py 1 --reps 50 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200
timeout 300 python tools/c1_probe.py 1 --reps 3 --set host_trace=1 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-260
