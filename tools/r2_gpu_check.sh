#!/bin/bash
# Round-2 first box lease: new device paths first, then the whole GPU suite, then timings.
#   gpurun --timeout 1500 -- 'bash tools/r2_gpu_check.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r2a}
mkdir -p "$OUT"
cd "$REPO"
export PYTHONUNBUFFERED=1
echo "== new paths =="; timeout 600 python -m pytest tests/test_gpu_r2_paths.py -q -m gpu 2>&1 | tail -40 | tee "$OUT/pytest_r2_paths.txt"
echo "== gpu suite =="; timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_r2_paths.py 2>&1 | tail -15 | tee "$OUT/pytest_gpu.txt"
echo "== trace of one C2 match =="; CMX_TRACE=1 timeout 120 python bench.py --steps 3 --warmup 3 --no-cpu-baseline 2>&1 | tail -8 | tee "$OUT/trace_c2.txt"
echo "== bench default =="; timeout 300 python bench.py --no-cpu-baseline 2> "$OUT/bench_default.err" | tail -1 | tee "$OUT/bench_default.json"
echo "== bench CMX_FUSED=0 =="; CMX_FUSED=0 timeout 300 python bench.py --no-cpu-baseline 2> "$OUT/bench_unfused.err" | tail -1 | tee "$OUT/bench_unfused.json"
echo "== C1 timings =="; timeout 300 python tools/time_configs.py c1 c1b 2>&1 | tee "$OUT/time_c1.txt"
echo "== C1 timings, legacy =="; CMX_RT2D_BULK=0 timeout 300 python tools/time_configs.py c1 c1b 2>&1 | tee "$OUT/time_c1_legacy.txt"
