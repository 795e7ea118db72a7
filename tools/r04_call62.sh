#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P1() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | grep "^C1" | cut -c17-50; }
echo "== copy kernels"; P1 256 512 1024 2048 --reps 25
echo "== copy commands"; P1 256 512 1024 2048 --reps 25 --set no_copy_kernels=1
