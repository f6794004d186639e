#!/usr/bin/env python3
"""kernel timeline of a few short commitments out of a rocprofv3 kernel trace (rocpd database) of tools/msm_size_probe.py:
usage: tools/small_timeline.py <results.db> [which occurrence of msm_small_accumulate to start at = 10] [kernels to print = 14]"""
import re
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x, grid_y from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "msm_small_acc" in r[0]]
i0 = idx[int(sys.argv[2]) if len(sys.argv) > 2 else 10]
t0 = rows[i0][1]
for r in rows[i0 - 1:i0 + (int(sys.argv[3]) if len(sys.argv) > 3 else 14)]:
    print("%9.1f us +%7.1f us  grid %6d x %d  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[4], re.sub(r"\(.*", "", r[0].replace("plk::", ""))[:44]))
