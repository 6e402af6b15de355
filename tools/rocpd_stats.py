#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max in microseconds) of a rocprofv3 run whose output is the
rocpd SQLite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db on this
ROCm), printed as CSV: the same columns as rocprofv3's kernel_stats.csv.

  python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/r02_x_kernel_stats.csv"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for name, calls, tot, avg, mn, mx in rows:
        print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (name, calls, tot, avg, 100.0 * tot / total, mn, mx))


if __name__ == "__main__":
    main(sys.argv[1])
