# C5 share of 32 distinct submaps: node placement variants.
mkdir -p gpurun_out/r2u
for af in 1 2 1 2; do
CMX_FAST3D_AFFINITY=$af timeout 200 python bench.py --config c5 --submaps 32 --steps 4 --warmup 2 --no-cpu-baseline --pmc-dir gpurun_out 2> gpurun_out/r2u/c5_aff$af.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('affinity', $af, d['ms_per_step'], c.get('device_ms_per_step'), c.get('found'), c.get('nodes_expanded_per_step'), r.get('kernel_ms'), r.get('frac'))" | tee -a gpurun_out/r2u/c5_affinity2.txt
done
CMX_TRACE=1 CMX_FAST3D_AFFINITY=2 timeout 200 python bench.py --config c5 --submaps 32 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "cmx trace" | tail -3 | tee -a gpurun_out/r2u/c5_affinity2.txt
CMX_FAST3D_AFFINITY=2 timeout 300 python -m pytest tests/test_gpu_3d.py -m gpu -x -q -p no:cacheprovider -k fast3d 2>&1 | grep -E "passed|failed" | tee -a gpurun_out/r2u/c5_affinity2.txt
