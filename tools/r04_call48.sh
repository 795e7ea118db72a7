#!/bin/bash
# every device test + the default bench line (bench with the event brackets off in the timed regions)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_final3
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $OUT/pytest_gpu.txt 2>&1 < /dev/null
grep -E "passed|failed|error" $OUT/pytest_gpu.txt | tail -3
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.txt | head -20
( time timeout 900 python bench.py ) > $OUT/bench_full.log 2> $OUT/bench_full.err < /dev/null
tail -3 $OUT/bench_full.err | cut -c1-300
tail -1 $OUT/bench_full.log | cut -c1-1500
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -5
