#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_r2_paths.py -m gpu -q -p no:cacheprovider -k "histogram or voxel or timing" ) 2>&1 | tail -4
timeout 200 python tools/family_probe.py 2>&1 | grep "histogram"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_h && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o kt -- python $REPO/tools/family_probe.py > /dev/null 2>&1
python $REPO/profiles/rocpd_summary.py $(find /tmp/prof_h -name '*.db' | head -1) $REPO/gpurun_out/r04_family_kernel_stats_v2.csv > /dev/null 2>&1
grep -i "slice\|histogram" $REPO/gpurun_out/r04_family_kernel_stats_v2.csv | sed 's/cmx::(anonymous namespace):://g' | awk -F'"' '{print substr($2,1,40), $3}' | cut -c1-120
