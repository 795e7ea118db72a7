#!/bin/bash
# BASELINE configs [2] and [4] at their FULL size on the one GPU of a gpurun box (the driver's
# scaling run shards them over eight): 512 distinct 400 x 400 submaps / 256 distinct 150^3
# submap pairs resident in one HBM, EVERY result of both checked against the reference by the
# bench's parity gate (--parity-submaps).  gpurun --timeout 1500 -- 'bash tools/full_configs.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=gpurun_out/full_configs; mkdir -p $OUT
export PYTHONUNBUFFERED=1
echo "== C3: one scan vs 512 submaps, parity on all 512"
timeout 900 python bench.py --config c3 --submaps 512 --parity-submaps 512 --steps 5 --warmup 2 --passes-per-step 1 --no-other --no-cpu-baseline --details $OUT/c3_512_details.json 2> $OUT/c3_512.err | tee $OUT/c3_512.json | cut -c1-1200
echo "== C5: one node vs 256 submaps, parity on all 256"
timeout 900 python bench.py --config c5 --submaps 256 --parity-submaps 256 --steps 3 --warmup 1 --passes-per-step 1 --no-other --no-cpu-baseline --details $OUT/c5_256_details.json 2> $OUT/c5_256.err | tee $OUT/c5_256.json | cut -c1-1200
for f in $OUT/c3_512.err $OUT/c5_256.err; do tail -n 3 $f; done
