# Round-2 experiment driver: full device tests + a host trace of the RT-2D batch.
mkdir -p gpurun_out/r2w
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r2w/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r2w/pytest_gpu.txt | tail -3
grep -B30 "Error\|assert" gpurun_out/r2w/pytest_gpu.txt | tail -60
CMX_HOST_TRACE=1 timeout 120 python bench.py --config c1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r2w/c1.json 2> gpurun_out/r2w/c1_host_trace.txt
grep "cmx host" gpurun_out/r2w/c1_host_trace.txt | tail -4
python - <<'P'
import json
d=json.loads(open('gpurun_out/r2w/c1.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config'].get('device_ms_per_step'), d['value'])
P
