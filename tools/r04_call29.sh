#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
timeout 300 python tools/timeline_probe.py c1b 2>&1 | grep -v amdgpu.ids | grep -A28 "C1 batch 512" | tail -64 | cut -c1-200
