#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P1() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | grep "^C1" | cut -c1-100; }
echo "== default"; P1 256 512 1024 2048 --reps 30
echo "== groups 1"; P1 256 512 1024 2048 --reps 30 --set rt2d_groups=1
echo "== groups 2"; P1 512 1024 2048 --reps 30 --set rt2d_groups=2
echo "== groups 1, equal parts 4"; P1 512 1024 2048 --reps 30 --set rt2d_groups=1 --set rt2d_parts=4
echo "== groups 1, equal parts 2"; P1 256 512 1024 2048 --reps 30 --set rt2d_groups=1 --set rt2d_parts=2
echo "== groups 1, equal parts 8"; P1 1024 2048 --reps 30 --set rt2d_groups=1 --set rt2d_parts=8
