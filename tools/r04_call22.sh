#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $OUT/r04_pytest_gpu_v2.txt 2>&1 < /dev/null
grep -E "passed|failed|error" $OUT/r04_pytest_gpu_v2.txt | tail -3
grep -E "^FAILED|^ERROR" $OUT/r04_pytest_gpu_v2.txt | head -20
( time timeout 900 python bench.py ) > $OUT/r04_bench_full_v2.log 2>&1 < /dev/null
grep '^{"metric' $OUT/r04_bench_full_v2.log | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
