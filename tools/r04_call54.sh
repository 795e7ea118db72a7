#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P1() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | grep "^C1" | cut -c1-110; }
echo "== default"; P1 64 128 256 --reps 40
for G in 1 2 3 6; do echo "== groups $G"; P1 64 128 256 --reps 40 --set rt2d_groups=$G; done
echo "== default"; P1 128 --reps 40
