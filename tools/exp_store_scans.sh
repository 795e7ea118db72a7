# Experiment driver (round 2): cells of surviving scans stored by the front end, XCD affinity.
mkdir -p gpurun_out/r2z
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r2z/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r2z/pytest_gpu.txt | tail -3
run() {  # label, env..., args
  label=$1; shift
  echo "== $label" >> gpurun_out/r2z/c3.txt
  env "$@" | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print(d['ms_per_step'], c.get('device_ms_per_step'), r['kernel'][:20], r.get('kernel_ms'), r.get('frac'), r.get('launches_per_step'), r.get('lookups_per_step'), r.get('nodes_per_step'))" >> gpurun_out/r2z/c3.txt
}
B="timeout 120 python bench.py --no-cpu-baseline --no-other"
run "c3x16 default"      CMX_X=0 $B --config c3 --submaps 16 --steps 5 --warmup 2 2>/dev/null
run "c3x16 affinity1"    CMX_XCD_AFFINITY=1 $B --config c3 --submaps 16 --steps 5 --warmup 2 2>/dev/null
run "c3x16 store0 aff0"  CMX_STORE_SCANS=0 CMX_XCD_AFFINITY=0 $B --config c3 --submaps 16 --steps 5 --warmup 2 2>/dev/null
run "c3x64 default"      CMX_X=0 $B --config c3 --submaps 64 --steps 3 --warmup 1 2>/dev/null
run "c3x64 affinity0"    CMX_XCD_AFFINITY=0 $B --config c3 --submaps 64 --steps 3 --warmup 1 2>/dev/null
run "c3x8 default"       CMX_X=0 $B --config c3 --submaps 8 --steps 5 --warmup 2 2>/dev/null
run "c3x8 affinity0"     CMX_XCD_AFFINITY=0 $B --config c3 --submaps 8 --steps 5 --warmup 2 2>/dev/null
run "c2"                 CMX_X=0 $B --steps 300 --warmup 100 2>/dev/null
run "c5x32"              CMX_X=0 $B --config c5 --submaps 32 --steps 4 --warmup 2 2>/dev/null
CMX_TRACE=1 timeout 120 python bench.py --config c3 --submaps 16 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "cmx trace" | tail -8 > gpurun_out/r2z/trace.txt
cat gpurun_out/r2z/c3.txt; tail -3 gpurun_out/r2z/trace.txt
