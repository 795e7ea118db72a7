#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export PYTHONUNBUFFERED=1
echo "== host trace of a 1024-match call"
timeout 300 python tools/c1_probe.py 1024 --reps 3 --set host_trace=1 2>&1 | grep "cmx host" | tail -14 | cut -c1-260
for P in 1 2 3 4; do echo "== rt2d_parts=$P"; timeout 300 python tools/c1_probe.py 1024 2048 --reps 25 --set rt2d_parts=$P 2>&1 | grep "^C1" | cut -c1-120; done
