#!/bin/bash
# Kernel trace of the staged C4 match (two matches), per-kernel statistics and every launch of the
# RT-3D kernels in start order.   gpurun -- 'bash tools/profile_c4_staged.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export CMX_SKIP_TORCH_IMPORT=1
export CMX_NO_REPORT=1      # clean kernel durations
PROFILE_TIMEOUT=100 bash "$REPO/tools/profile_cmd.sh" r03s_c4 "python tools/time_rt3d.py 1" < /dev/null
DB=$(find /tmp/prof_r03s_c4_kt -name '*.db' | head -1)
python "$REPO/profiles/rocpd_summary.py" "$DB" --calls Rt3D > "$REPO/gpurun_out/r03s_c4_launches.txt"
tail -40 "$REPO/gpurun_out/r03s_c4_launches.txt"
