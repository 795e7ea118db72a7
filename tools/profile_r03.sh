#!/bin/bash
# Round-3 profiles (bounded: every pass under its own timeout, stdin closed).
#   gpurun --timeout 900 -- 'bash tools/profile_r03.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PROFILE_TIMEOUT=100
P="bash tools/profile_cmd.sh"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
$P r03    "python bench.py --steps 4 --warmup 1 --passes-per-step 64 --no-cpu-baseline --no-other" FETCH_SIZE WRITE_SIZE < /dev/null
$P r03_c1 "python bench.py --config c1 --steps 6 --warmup 2 --no-cpu-baseline" FETCH_SIZE WRITE_SIZE "$SQ" < /dev/null
$P r03_c3 "python bench.py --config c3 --submaps 16 --steps 3 --warmup 1 --no-cpu-baseline" < /dev/null
$P r03_c5 "python bench.py --config c5 --submaps 32 --steps 3 --warmup 1 --no-cpu-baseline" < /dev/null
$P r03_c4 "python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline" FETCH_SIZE WRITE_SIZE < /dev/null
ls -la gpurun_out/r03*kernel_stats.csv
