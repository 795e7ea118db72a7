"""The last dispatches of a rocprofv3 kernel trace (rocpd sqlite) as a timeline: start, end,
queue and name -- how the kernels of a call's parts overlap on the device.
   python tools/dispatch_timeline.py results.db [count]"""
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
count = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
sel = "name, start, end" + (f", {qcol}" if qcol else "") + (", grid_size, workgroup_size" if "grid_size" in cols else "")
rows = con.execute(f"select {sel} from kernels order by start").fetchall()
rows = rows[-count:]
t0 = rows[0][1]
for r in rows:
    import re
    m = re.search(r"::(\w+)(<[^>]*>)?\(", r[0])
    short = (m.group(1) + (m.group(2) or "")) if m else r[0][:28]
    extra = " ".join(str(x) for x in r[3:])
    print(f"{(r[1] - t0) / 1e3:9.1f} -> {(r[2] - t0) / 1e3:9.1f} us ({(r[2] - r[1]) / 1e3:7.1f})  {short:28s} {extra}")
