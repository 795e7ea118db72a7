#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-260; }
echo "== 128 host trace"; P 128 --reps 6 --set host_trace=1 | tail -40
echo "== 1024 host trace"; P 1024 --reps 3 --set host_trace=1 | tail -60
