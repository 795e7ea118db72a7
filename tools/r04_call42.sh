#!/bin/bash
# exact, wave-parallel rotational histogram; fused front-end block size under 8 host threads
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/r04_call42
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 600 python -m pytest tests/test_gpu_r2_paths.py -m gpu -q -p no:cacheprovider -k "histogram or voxel" ) > $OUT/pytest.txt 2>&1
tail -25 $OUT/pytest.txt | cut -c1-250
timeout 200 python tools/family_probe.py 2>&1 | grep "histogram\|voxel"
P2() { timeout 200 python tools/c2_probe.py "$@" 2>&1 | grep "^\[" | cut -c1-200; }
P2 --no-c3 --set fast2d_fused_threads=128
P2 --no-c3 --set fast2d_fused_threads=64
P2 --no-c3 --set fast2d_fused_threads=256
P2 --no-c3
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_h && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o kt -- python $REPO/tools/family_probe.py > /dev/null 2>&1
python $REPO/profiles/rocpd_summary.py $(find /tmp/prof_h -name '*.db' | head -1) $REPO/$OUT/family_kernel_stats.csv > /dev/null 2>&1
grep -i "slice\|histogram\|RadixSort\|Draw\|Voxel" $REPO/$OUT/family_kernel_stats.csv | sed 's/cmx::(anonymous namespace):://g' | awk -F'"' '{print substr($2,1,60), $3}' | cut -c1-150 | head -20
