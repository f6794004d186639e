#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes its kernel trace as a rocpd SQLite database unless --output-format csv is given; this prints
the same per-kernel summary as the *_kernel_stats.csv of --stats from such a database.
usage: tools/rocpd_stats.py <results.db> [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                  "group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
for name, calls, tot, avg, mn, mx in rows:
    lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (name.replace('"', "'"), calls, tot, avg, 100.0 * tot / total, mn, mx))
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
else:
    sys.stdout.write(out)
