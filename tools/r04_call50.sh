#!/bin/bash
# counters of the SHIPPED kernels: SQ passes for C4's three tile kernels, cache passes for C5's expansion
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
export PROFILE_TIMEOUT=150
bash tools/profile_cmd.sh r04s_c4 "python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-other" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
  "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" > /dev/null 2>&1
bash tools/profile_cmd.sh r04s_c5 "python bench.py --config c5 --submaps 32 --steps 3 --warmup 1 --no-cpu-baseline --no-other" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES" > /dev/null 2>&1
bash tools/profile_cmd.sh r04s_c2 "python bench.py --concurrency 1 --steps 4 --warmup 1 --passes-per-step 64 --no-cpu-baseline --no-other" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" > /dev/null 2>&1
ls gpurun_out/r04s_* | head -30
for f in gpurun_out/r04s_c4_pmc_sq_wave*.csv gpurun_out/r04s_c5_pmc_tcp*.csv gpurun_out/r04s_c2_pmc_*.csv; do echo "== $f"; grep "Rt3DTile\|Expand3D\|PrepScoreFused" "$f" | sed 's/cmx::(anonymous namespace):://g' | awk -F'"' '{print substr($2,1,40), $3}' | cut -c1-130; done
