#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P2() { timeout 200 python tools/c2_probe.py "$@" 2>&1 | grep "^\[\|Error\|error" | cut -c1-200; }
P2 --no-c3
P2 --no-c3 --threads 4
P2 --no-c3 --threads 16
