#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
timeout 300 python tools/time_rt3d.py 1 2>&1 | grep -v amdgpu.ids | cut -c1-900 | tail -40
