#!/bin/bash
# Round 5: the whole GPU suite, smoke, the default bench (compact line), as the driver runs them.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=gpurun_out/r05_call11; mkdir -p $OUT
export PYTHONUNBUFFERED=1
echo "== gpu tests"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== default bench"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "rc $?"; wc -c $OUT/bench.json; cat $OUT/bench.json; tail -5 $OUT/bench.err
cp gpurun_out/bench_details.json $OUT/bench_details.json
