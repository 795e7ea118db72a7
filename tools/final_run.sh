# Round-end validation on the GPU box: device tests, profiles of every config, the full bench line.
TAG=${1:-r02e}
mkdir -p gpurun_out/final
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/final/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" gpurun_out/final/pytest_gpu.txt | tail -3
PROFILE_TIMEOUT=100 bash tools/profile_all.sh $TAG > gpurun_out/final/profile_all.log 2>&1
tail -12 gpurun_out/final/profile_all.log
timeout 400 python bench.py --pmc-dir gpurun_out --pmc-tag $TAG > gpurun_out/final/bench_full.json 2> gpurun_out/final/bench_full.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/final/bench_full.json').read().strip().splitlines()[-1])
r=d['roofline']; print('c2', d['ms_per_step'], d['value'], r['frac'], r.get('traffic'), d['cpu_baseline']['value'])
for k,v in d['config']['other'].items():
    rf=v.get('roofline') or {}
    print(k, v.get('ms_per_step'), v.get('device_ms_per_step'), v.get('candidates_per_s'), v.get('matches_per_s'), v.get('error'), (rf.get('kernel') or '')[:18], rf.get('frac'), rf.get('traffic'))
P
