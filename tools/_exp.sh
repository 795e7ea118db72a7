mkdir -p gpurun_out/r06
for b in 512 384 256; do
python tools/c2m_probe.py --reps 30 --set fast2d_queue_blocks=$b 2>&1 | grep -v "amdgpu.ids"
done > gpurun_out/r06/c2m_queue8.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_queue.py tests/test_gpu_2d.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06/t2d.txt
