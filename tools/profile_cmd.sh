#!/bin/bash
# rocprofv3 of an arbitrary command on the GPU box: kernel-trace statistics, then one PMC
# pass per counter group (separate runs: gpurun refuses --pmc together with tracing domains
# other than --kernel-trace).  CSV summaries go to gpurun_out/<tag>_*.csv.
#   bash tools/profile_cmd.sh <tag> "<command>" ["<counter group>" ...]
# e.g. bash tools/profile_cmd.sh r02_c1b "python tools/timeline_probe.py c1b" FETCH_SIZE WRITE_SIZE \
#          "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"
set -u
TAG=$1; CMD=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_${TAG}_kt
# A pass that faults at start-up leaves rocprofv3 waiting for ever: bounded, and the counter
# passes are skipped when the trace pass did not finish.
LIMIT=${PROFILE_TIMEOUT:-120}
( cd "$REPO" && timeout -k 5 $LIMIT rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_kt -o kt -- $CMD ) > "$OUT/${TAG}_kt.log" 2>&1 \
  || { echo "[profile_cmd] $TAG: trace pass failed or timed out (see ${TAG}_kt.log); skipping the counter passes"; exit 1; }
python "$REPO/profiles/rocpd_summary.py" $(find /tmp/prof_${TAG}_kt -name '*.db' | head -1) "$OUT/${TAG}_kernel_stats.csv" > /dev/null
i=0
for GROUP in "$@"; do
  i=$((i + 1))
  NAME=$(echo "$GROUP" | tr ' ' '+' | tr A-Z a-z | cut -c1-60)
  rm -rf /tmp/prof_${TAG}_$i
  ( cd "$REPO" && timeout -k 5 $LIMIT rocprofv3 --kernel-trace --pmc $GROUP -d /tmp/prof_${TAG}_$i -o pmc -- $CMD ) > "$OUT/${TAG}_pmc_$i.log" 2>&1 \
    || { echo "[profile_cmd] $TAG: counter pass '$GROUP' failed or timed out"; continue; }
  python "$REPO/profiles/rocpd_pmc_summary.py" $(find /tmp/prof_${TAG}_$i -name '*.db' | head -1) "$OUT/${TAG}_pmc_${NAME}.csv" > /dev/null
done
head -6 "$OUT/${TAG}_kernel_stats.csv" | cut -c1-200
