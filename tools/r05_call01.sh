#!/bin/bash
# Round 5, call 1: the pending front-end step patch A/B'd against round 4's library on one box,
# its tests, then the default bench (compact line + parity gate) and the C1 call sizes.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=gpurun_out/r05_call01; mkdir -p $OUT
export PYTHONUNBUFFERED=1
bash tools/ab.sh r05_call01 "python tools/c2_probe.py --no-c3 2>&1 | grep '^\['"
echo "== front-end tests (patched library)"
timeout 600 python -m pytest tests/test_gpu_2d.py tests/test_gpu_r2_paths.py -m gpu -q -p no:cacheprovider -k "prepare or front_ends or c3_share" 2>&1 | tail -3 | tee $OUT/front_tests.txt
echo "== default bench"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "rc $?"; wc -c $OUT/bench.json; cat $OUT/bench.json; tail -5 $OUT/bench.err
cp gpurun_out/bench_details.json $OUT/bench_details.json
echo "== C1 call sizes"
timeout 300 python tools/c1_probe.py 1 128 1024 --reps 25 2>&1 | grep "^C1" | cut -c1-160 | tee $OUT/c1_probe.txt
