#!/bin/bash
# (record of an experiment: the debug switch it sets was removed together with the variant that lost -- DESIGN.md 5.3)
# tile workgroups that exit after N items (CUs free up for the other parts' finish kernels)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P1() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | grep "^C1" | cut -c1-120; }
echo "== persistent"; P1 128 512 1024 2048 --reps 30
for N in 1 2 4; do echo "== items per wg $N"; P1 128 512 1024 2048 --reps 30 --set rt2d_items_per_wg=$N; done
echo "== persistent"; P1 1024 --reps 30
echo "== trace N=1"; timeout 300 python tools/c1_probe.py 1024 --reps 2 --set host_trace=1 --set rt2d_items_per_wg=1 2>&1 | grep "collected\|batch(" | tail -5 | cut -c1-200
