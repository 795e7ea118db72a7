mkdir -p gpurun_out/r06
for b in "fast2d_queue_lost=2" "fast2d_queue_lost=1" "fast2d_queue_lost=4"  ; do
a=""; for kv in $b; do a="$a --set $kv"; done
python tools/c2m_probe.py --reps 20 --trace $a 2>&1 | grep -v "amdgpu.ids\|fused front\|problem 0"
done > gpurun_out/r06/c2m_queue6.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_2d.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06/t2d.txt
