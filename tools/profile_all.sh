#!/bin/bash
# Round profiles: rocprofv3 kernel-trace statistics + FETCH_SIZE / WRITE_SIZE passes for every
# BASELINE config (the bench commands themselves) and for the kernel families the bench does not
# time.  Summaries land in gpurun_out/<tag>*.csv; copy what is to be judged into profiles/.
#   gpurun --timeout 1500 -- 'bash tools/profile_all.sh r06'
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
P="bash tools/profile_cmd.sh"
# (--no-parity: the device-vs-reference gate of bench.py is CPU time, repeated by every pass)
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
CACHE="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
# the headline: C2 from 16 host threads (16 hardware queues) over 8 DIFFERENT scans (round 6; rounds 2 - 5 profiled one
# easy scan), then the same single-stream
$P ${TAG}       "python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other --no-parity" FETCH_SIZE WRITE_SIZE "$SQ" "$CACHE"
$P ${TAG}_c2single "python bench.py --steps 12 --warmup 3 --concurrency 1 --no-cpu-baseline --no-other --no-parity" FETCH_SIZE WRITE_SIZE
LDS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
$P ${TAG}_c1    "python bench.py --config c1 --steps 6 --warmup 2 --no-cpu-baseline --no-parity" FETCH_SIZE WRITE_SIZE "$LDS"
# (round 5: 1024 matches per call, the bound kernel's own shape -- two workgroups per CU)
$P ${TAG}_c1b1024 "python bench.py --config c1 --matches 1024 --steps 6 --warmup 2 --no-cpu-baseline --no-parity" FETCH_SIZE WRITE_SIZE "$LDS"
$P ${TAG}_c3    "python bench.py --config c3 --submaps 16 --steps 3 --warmup 1 --no-cpu-baseline --no-parity" FETCH_SIZE WRITE_SIZE "$SQ" "$CACHE"
$P ${TAG}_c4    "python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-parity" FETCH_SIZE WRITE_SIZE
$P ${TAG}_c5    "python bench.py --config c5 --submaps 32 --steps 3 --warmup 1 --no-cpu-baseline --no-parity" FETCH_SIZE WRITE_SIZE "$SQ" "$CACHE"
$P ${TAG}_other "python tools/family_probe.py" FETCH_SIZE WRITE_SIZE
ls -la gpurun_out/${TAG}*kernel_stats.csv
