#!/bin/bash
# First gpurun call of a round: everything needed to decide where the round's GPU minutes go,
# in one box lease (~10 min).  Logs land in gpurun_out/round_start/.
#   gpurun --timeout 1200 -- 'bash tools/round_start.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/round_start
mkdir -p "$OUT"
cd "$REPO"
export PYTHONUNBUFFERED=1

echo "== gpu tests ==";          timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee "$OUT/pytest_gpu.txt"
echo "== device vs oracle, random odd-shaped problems ==";  timeout 300 python tools/gpu_fuzz.py 150 1 2>&1 | tail -20 | tee "$OUT/gpu_fuzz.txt"
echo "== bench (default) ==";    timeout 300 python bench.py 2> "$OUT/bench_default.err" | tail -1 | tee "$OUT/bench_default.json"
for T in 2 4 8; do               # host threads issuing independent searches (bench.py --concurrency)
  echo "== bench --concurrency $T =="
  timeout 200 python bench.py --concurrency $T --no-cpu-baseline 2> "$OUT/bench_c$T.err" | tail -1 | tee "$OUT/bench_c$T.json"
done
for B in 16 64; do               # ConstraintBuilder batches (BASELINE config[2] per-GPU share)
  echo "== bench --submaps $B =="
  timeout 300 python bench.py --submaps $B --steps 10 --warmup 3 --no-cpu-baseline 2> "$OUT/bench_b$B.err" | tail -1 | tee "$OUT/bench_b$B.json"
done
echo "== other configs ==";      timeout 600 python tools/time_configs.py c1 c1b c4 c5 --cpu 2>&1 | tee "$OUT/time_configs.txt"
echo "== hbm copy ==";           timeout 120 python tools/hbm_copy_bench.py 4 2>&1 | tee "$OUT/hbm_copy.txt"
echo "== kernel trace: single match and batch 64 =="
bash tools/profile_bench.sh round_start > "$OUT/profile_single.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b64
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b64 -o kt -- \
  python "$REPO/bench.py" --submaps 64 --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/kt_b64.log" 2>&1
python "$REPO/profiles/rocpd_summary.py" $(find /tmp/prof_b64 -name '*.db' | head -1) "$OUT/batch64_kernel_stats.csv" > /dev/null
head -12 "$OUT/batch64_kernel_stats.csv" | cut -c1-160
