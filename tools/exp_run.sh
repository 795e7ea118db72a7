mkdir -p gpurun_out/r2t
timeout 900 python -m pytest tests/test_dropin.py tests/test_gpu_3d.py tests/test_gpu_r2_paths.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r2t/pytest.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r2t/pytest.txt | tail -3
grep -B40 "^E  " gpurun_out/r2t/pytest.txt | tail -80
