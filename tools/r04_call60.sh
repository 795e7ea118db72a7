#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
timeout 300 python tools/probes/c1_after_threads.py 512 1024 2048 2>&1 | grep "^C1"
timeout 300 python tools/c1_probe.py 1024 --reps 30 2>&1 | grep "^C1" | cut -c17-60
( timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "rt2d or c1 or resident" ) 2>&1 | tail -2
