#!/bin/bash
# Round-5 A/B harness: the same probes against two builds of the library in ONE box lease
# (boxes differ by up to 40 % on the HBM-bound lines, so only same-box pairs mean anything).
#   gpurun --timeout 900 -- 'bash tools/ab.sh <tag> "<probe cmd>" ...'
# Libraries: cartographer_amd/lib/base_r04/ (the round-3 tree) vs cartographer_amd/lib/ (this tree).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
BASE=$REPO/cartographer_amd/lib/base_r04/libcartographer_mi355x.so
NEW=$REPO/cartographer_amd/lib/libcartographer_mi355x.so
export PYTHONUNBUFFERED=1
i=0
for CMD in "$@"; do
  i=$((i + 1))
  for ROUND in 1 2; do
    for WHICH in base new; do
      if [ $WHICH = base ]; then SO=$BASE; else SO=$NEW; fi
      echo "== [$i.$ROUND] $WHICH: $CMD" | tee -a "$OUT/ab.txt"
      CMX_SO_PATH=$SO timeout 300 bash -c "$CMD" 2>&1 | grep -v "^$" | tail -12 | tee -a "$OUT/ab.txt"
    done
  done
done
