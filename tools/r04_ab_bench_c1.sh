#!/bin/bash
# Same-box A/B of the C1 bench legs as the driver runs them (round-3 library vs this tree).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
ONE='import json,sys; o=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print("ms_per_step %.4f value %.4e device_ms %s" % (o["ms_per_step"], o["value"], o["config"].get("device_ms_per_step")))'
bash tools/ab_r04.sh r04_ab_bench_c1 \
  "python bench.py --config c1 --steps 200 --warmup 20 --no-cpu-baseline --no-other | python -c '$ONE'" \
  "python bench.py --config c1 --matches 1024 --steps 50 --warmup 10 --no-cpu-baseline --no-other | python -c '$ONE'" \
  "python bench.py --config c1 --matches 1 --steps 200 --warmup 20 --no-cpu-baseline --no-other | python -c '$ONE'" \
  "python tools/c1_probe.py 128"
timeout 600 python tools/c5_seed_ceiling.py 32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_c5_seed_ceiling.txt
