#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 (ROCm 7.2) rocpd sqlite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes NAME_results.db).
Usage: rocpd_summary.py results.db [out.csv]"""
import sqlite3
import sys


def calls(con, pattern):
    """Every launch whose name contains `pattern`, in start order: duration in microseconds."""
    rows = con.execute("select name, start, duration from kernels where name like ? order by start",
                       (f"%{pattern}%",)).fetchall()
    t0 = rows[0][1] if rows else 0
    for name, start, duration in rows:
        short = name.split("(")[0].split("::")[-1]
        print(f"{(start - t0) / 1e3:12.1f} us  {duration / 1e3:10.1f} us  {short}")


def main():
    con = sqlite3.connect(sys.argv[1])
    if len(sys.argv) > 3 and sys.argv[2] == "--calls":
        return calls(con, sys.argv[3])
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPR,SGPR,LDS"]
    for r in rows:
        lines.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]},{r[5]},{100.0 * r[2] / total:.2f},"
                     f"{r[6]},{r[7]},{r[8]}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
