#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
PROFILE_TIMEOUT=150 bash tools/profile_cmd.sh r04_call39/c1b "python tools/c1_probe.py 1024 --reps 3 --set rt2d_parts=1" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
  "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" > /dev/null 2>&1
for f in gpurun_out/r04_call39/c1b_pmc_*.csv; do echo "== $f"; grep "Rt2DTileKernel\|Rt2DFinish" "$f" | sed 's/cmx::(anonymous namespace):://g' | awk -F'"' '{print substr($2,1,34), $3}' | cut -c1-140; done
grep "Rt2D" gpurun_out/r04_call39/c1b_kernel_stats.csv | sed 's/cmx::(anonymous namespace):://g' | awk -F'"' '{print substr($2,1,34), $3}' | cut -c1-140
