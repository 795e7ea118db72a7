#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-110; }
for parts in 1 2; do echo "== 128: parts $parts"; P 128 --reps 40 --set rt2d_parts=$parts --set rt2d_groups=4; done
echo "== 128: parts 2 split"; P 128 --reps 40 --set rt2d_parts=2 --set rt2d_groups=4 --set rt2d_split=1
for parts in 1 2 3 4; do echo "== 1024: parts $parts"; P 1024 --reps 30 --set rt2d_parts=$parts; done
for parts in 2 3 4; do echo "== 1024: parts $parts split"; P 1024 --reps 30 --set rt2d_parts=$parts --set rt2d_split=1; done
