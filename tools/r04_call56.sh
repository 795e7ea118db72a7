#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P1() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | grep "^C1" | cut -c17-60; }
for PARTS in 1 2 3; do for G in 1 2 3; do echo "== parts $PARTS groups $G"; P1 256 384 512 768 1024 2048 --reps 25 --set rt2d_groups=$G --set rt2d_parts=$PARTS; done; done
