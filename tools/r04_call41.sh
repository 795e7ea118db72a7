#!/bin/bash
# hardware queues: 8 host threads / 8 batch parts over the runtime's default 4 queues vs 8 / 16
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P2() { timeout 200 python tools/c2_probe.py "$@" 2>&1 | grep "^\[" | cut -c1-200; }
P1() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-120; }
for Q in 4 8 16; do
  export GPU_MAX_HW_QUEUES=$Q
  echo "== GPU_MAX_HW_QUEUES=$Q"
  P2 --no-c3
  P2 --no-c3 --threads 16
  P1 128 1024 --reps 30
  if [ $Q != 4 ]; then P1 1024 --reps 30 --set rt2d_parts=8; fi
done
