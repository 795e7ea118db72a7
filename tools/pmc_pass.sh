#!/bin/bash
# One rocprofv3 PMC pass over the default bench: bash tools/pmc_pass.sh <tag> "<counters>" [bench flags]
set -u
TAG=$1; COUNTERS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 200 rocprofv3 --kernel-trace --pmc $COUNTERS -d /tmp/prof_$TAG -o pmc -- \
  python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > "$OUT/${TAG}.log" 2>&1
python "$REPO/profiles/rocpd_pmc_summary.py" $(find /tmp/prof_$TAG -name '*.db' | head -1) "$OUT/${TAG}.csv" > /dev/null
grep -h "ScoreCoarsePlanes\|ExpandWave" "$OUT/${TAG}.csv" | sed 's/"[^"]*::\([A-Za-z]*Kernel\)[^"]*"/\1/' 
