mkdir -p gpurun_out/final3
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/final3/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" gpurun_out/final3/pytest_gpu.txt | tail -3
grep -B30 "^E  " gpurun_out/final3/pytest_gpu.txt | tail -40
for cfg in "--config c3 --submaps 16 --steps 5 --warmup 2" "--steps 400 --warmup 150 --no-other"; do
timeout 120 python bench.py $cfg --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print(c['workload'][:3], d['ms_per_step'], c.get('device_ms_per_step'), r['kernel'][:16], r.get('kernel_ms'), r.get('frac'), r.get('lookups_per_step'))" | tee -a gpurun_out/final3/bench.txt
done
