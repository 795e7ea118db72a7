#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export PYTHONUNBUFFERED=1
timeout 20 python -m pytest tests/test_gpu_2d.py tests/test_gpu_r2_paths.py -m gpu -q -p no:cacheprovider -x -k "prepare or front_ends or c3_share" 2>&1 | tail -2
timeout 12 python tools/c2_probe.py --no-c3 --threads 1 2>&1 | grep "^\[" | cut -c1-150
