#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r04_call28
cd /tmp && export TMPDIR=/tmp
for B in 1024 128; do
rm -rf /tmp/prof_kt
( cd "$REPO" && timeout -k 5 200 rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt -- python tools/c1_probe.py $B --reps 2 ) > $REPO/gpurun_out/r04_call28/kt_$B.log 2>&1
DB=$(find /tmp/prof_kt -name '*.db' | head -1)
echo "== batch $B: last launches (start us, duration us, name)"
python - "$DB" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, duration from kernels order by start").fetchall()
rows = rows[-34:] if len(rows) > 34 else rows
t0 = rows[0][1]
for name, start, dur in rows:
    short = name.split("(")[0].split("::")[-1][:40]
    print(f"{(start - t0) / 1e3:10.1f} {dur / 1e3:8.1f} -> {(start - t0 + dur) / 1e3:8.1f}  {short}")
PY
done
