#!/bin/bash
# Round-3 closing run on the GPU box: every device test, the bench line (all configs + CPU
# baselines), the kernel trace of the staged C4 match.
#   gpurun --timeout 600 -- 'bash tools/final_r03.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
mkdir -p gpurun_out/final
( time timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/final/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" gpurun_out/final/pytest_gpu.txt | tail -3
( time timeout 300 python bench.py --steps 100 --warmup 20 ) > gpurun_out/final/bench_full.json 2> gpurun_out/final/bench_full.err
tail -c 1500 gpurun_out/final/bench_full.json
tail -3 gpurun_out/final/bench_full.err
bash tools/profile_c4_staged.sh > gpurun_out/final/profile_c4.log 2>&1
head -8 gpurun_out/r03s_c4_kernel_stats.csv | cut -c1-160
