#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
P1() { timeout 300 python tools/c1_probe.py "$@" 2>&1 | grep -v amdgpu.ids | grep "^C1" | cut -c17-110; }
P1 1024 --reps 30
timeout 300 python bench.py --config c1 --matches 1024 --steps 30 --warmup 5 --no-cpu-baseline --no-other 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench --config c1 --matches 1024: ms_per_step', d['ms_per_step'], 'passes', d['config'].get('passes_per_step'), 'ms_per_pass', d['config'].get('ms_per_pass'), 'value', d['value'])"
P1 1024 --reps 30
