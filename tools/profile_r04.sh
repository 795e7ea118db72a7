#!/bin/bash
# Round-4 evidence run: every device test, the default bench line, then rocprofv3 passes per
# config (kernel-trace statistics + FETCH_SIZE / WRITE_SIZE counter passes, each its own run).
#   gpurun --timeout 2400 -- 'bash tools/profile_r04.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_final
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $OUT/pytest_gpu.txt 2>&1 < /dev/null
grep -E "passed|failed|error" $OUT/pytest_gpu.txt | tail -3
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.txt | head -20
( time timeout 900 python bench.py ) > $OUT/bench_full.log 2>&1 < /dev/null
tail -1 $OUT/bench_full.log | cut -c1-600
export PROFILE_TIMEOUT=150
P="bash tools/profile_cmd.sh"
$P r04    "python bench.py --steps 4 --warmup 1 --passes-per-step 64 --no-cpu-baseline --no-other" FETCH_SIZE WRITE_SIZE < /dev/null
$P r04_c1 "python bench.py --config c1 --steps 6 --warmup 2 --no-cpu-baseline --no-other" FETCH_SIZE WRITE_SIZE < /dev/null
$P r04_c3 "python bench.py --config c3 --submaps 16 --steps 3 --warmup 1 --no-cpu-baseline --no-other" FETCH_SIZE WRITE_SIZE < /dev/null
$P r04_c5 "python bench.py --config c5 --submaps 32 --steps 3 --warmup 1 --no-cpu-baseline --no-other" FETCH_SIZE WRITE_SIZE < /dev/null
$P r04_c4 "python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-other" FETCH_SIZE WRITE_SIZE < /dev/null
ls gpurun_out/r04*kernel_stats.csv gpurun_out/r04*pmc*.csv | wc -l
