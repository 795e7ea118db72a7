#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P1() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | grep "^C1" | cut -c17-50; }
for PARTS in 2 3 4; do echo "== parts $PARTS, share"; P1 256 384 512 768 1024 2048 --reps 25 --set rt2d_parts=$PARTS --set rt2d_grid_share=1; done
echo "== parts 2, share, groups 1"; P1 256 384 512 768 1024 2048 --reps 25 --set rt2d_parts=2 --set rt2d_grid_share=1 --set rt2d_groups=1
echo "== parts 2, no share"; P1 256 384 512 768 1024 2048 --reps 25 --set rt2d_parts=2
echo "== default"; P1 256 384 512 768 1024 2048 --reps 25
