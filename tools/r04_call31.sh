#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-110; }
for parts in 1 2; do for g in 8 4 2; do echo "== 128: parts $parts groups<=$g"; P 128 --reps 40 --set rt2d_parts=$parts --set rt2d_groups=$g; done; done
for parts in 2 4 8; do for g in 4 2 1; do echo "== 1024: parts $parts groups<=$g"; P 1024 --reps 30 --set rt2d_parts=$parts --set rt2d_groups=$g; done; done
echo "== 1 match"; P 1 --reps 50; P 1 --reps 50 --set rt2d_groups=27;  P 1 --reps 50 --set rt2d_groups=14
