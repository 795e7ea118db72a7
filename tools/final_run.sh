#!/bin/bash
# The round's last GPU call: what the driver runs at round end (GPU suite, smoke, the default bench),
# the rocprofv3 passes of every config (tools/profile_all.sh -> gpurun_out/<tag>_*.csv: copy into
# profiles/), and the default bench once more with those passes in place.
#   gpurun --timeout 2400 -- 'bash tools/final_run.sh r06'
set -u
TAG=${1:-r06}
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=gpurun_out/final_$TAG; mkdir -p $OUT
export PYTHONUNBUFFERED=1
echo "== gpu tests"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
echo "== profiles"; bash tools/profile_all.sh $TAG 2>&1 | tail -3
cp gpurun_out/${TAG}*_kernel_stats.csv gpurun_out/${TAG}*_pmc_*.csv profiles/ 2>/dev/null
echo "== default bench (as the driver runs it)"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "rc $?"; wc -c $OUT/bench.json; cat $OUT/bench.json; tail -n 3 $OUT/bench.err
cp gpurun_out/bench_details.json $OUT/bench_details.json
echo "== c1 1024 as the timed config"
timeout 600 python bench.py --config c1 --matches 1024 --steps 20 --warmup 5 --details $OUT/bench_c1b1024_details.json 2>/dev/null | tee $OUT/bench_c1b1024.json | cut -c1-600
