#!/usr/bin/env python3
"""Per-kernel mean of one PMC counter from a rocprofv3 rocpd database
(`rocprofv3 --pmc FETCH_SIZE --kernel-trace -d DIR -o NAME -- cmd`).
FETCH_SIZE / WRITE_SIZE are in KiB.  Usage: rocpd_pmc_summary.py results.db [out.csv]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute(
        "select kernel_name, counter_name, count(*), avg(value), min(value), max(value), "
        "avg(duration) from counters_collection group by kernel_name, counter_name "
        "order by avg(value) * count(*) desc").fetchall()
    lines = ["Kernel,Counter,Dispatches,MeanValue,MinValue,MaxValue,MeanDurationNs"]
    for r in rows:
        lines.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.3f},{r[4]:.3f},{r[5]:.3f},{r[6]:.1f}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
