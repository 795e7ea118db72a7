#!/bin/bash
# kernel trace of the 1024-match C1 call under the equal-parts schedule
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
export PROFILE_TIMEOUT=120
bash tools/profile_cmd.sh r04_c1b1024_v2 "python bench.py --config c1 --matches 1024 --steps 6 --warmup 2 --no-cpu-baseline --no-other" > /dev/null 2>&1
head -8 gpurun_out/r04_c1b1024_v2_kernel_stats.csv | sed 's/cmx::(anonymous namespace):://g' | awk -F'"' '{print substr($2,1,40), $3}' | cut -c1-130
