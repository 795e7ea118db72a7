#!/bin/bash
# where a block of the fused front end spends its time: in-kernel timeline + instruction-mix counters
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
timeout 200 python tools/timeline_probe.py c2 2>&1 | grep -v amdgpu.ids | tail -30 | cut -c1-220
export PROFILE_TIMEOUT=150
bash tools/profile_cmd.sh r04t_c2 "python bench.py --concurrency 1 --steps 4 --warmup 1 --passes-per-step 64 --no-cpu-baseline --no-other" \
  "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
  "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_WAVE_CYCLES" > /dev/null 2>&1
for f in gpurun_out/r04t_c2_pmc_*.csv; do echo "== $f"; grep "PrepScoreFused" "$f" | sed 's/cmx::(anonymous namespace):://g' | awk -F'"' '{print substr($2,1,30), $3}' | cut -c1-120; done
