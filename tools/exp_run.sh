TAG=r02e
P="bash tools/profile_cmd.sh"
export PROFILE_TIMEOUT=100
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
CACHE="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
$P ${TAG}_c5 "python bench.py --config c5 --submaps 32 --steps 3 --warmup 1 --no-cpu-baseline" FETCH_SIZE WRITE_SIZE "$SQ" "$CACHE"
$P ${TAG}    "python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other" FETCH_SIZE WRITE_SIZE
ls -la gpurun_out/${TAG}_c5* gpurun_out/${TAG}_kernel_stats.csv gpurun_out/${TAG}_pmc*
