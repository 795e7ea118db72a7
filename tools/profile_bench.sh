#!/bin/bash
# Profiles the default bench line on the GPU box: kernel-trace statistics plus the two PMC
# passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Summaries go to gpurun_out/.
# Usage (through gpurun): bash tools/profile_bench.sh <tag> [extra bench.py flags]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 12 --warmup 3 --no-cpu-baseline $*"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $CMD > "$OUT/${TAG}_kt.log" 2>&1
python "$REPO/profiles/rocpd_summary.py" $(find /tmp/prof_kt -name '*.db' | head -1) "$OUT/${TAG}_kernel_stats.csv" > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_$C -o pmc -- $CMD > "$OUT/${TAG}_pmc_$C.log" 2>&1
  python "$REPO/profiles/rocpd_pmc_summary.py" $(find /tmp/prof_$C -name '*.db' | head -1) \
      "$OUT/${TAG}_pmc_$(echo $C | tr A-Z a-z).csv" > /dev/null
done
head -8 "$OUT/${TAG}_kernel_stats.csv"
grep -h ScoreCoarsePlanes "$OUT"/${TAG}_pmc_*.csv
