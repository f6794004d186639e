#!/usr/bin/env python3
"""timeline of the LAST proof in a rocprofv3 kernel trace of tools/prove_probe.py (rocpd database): every kernel in start
order with its stream, start offset, duration and the idle gap of the whole GPU before it — where the exposed latency is.
usage: tools/timeline.py <results.db> [n_proves_in_trace=7]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, stream_id, grid_x, workgroup_x from kernels order by start").fetchall()
# a proof starts at the k_witness / first kernel after a long host gap: split on gaps > 1 ms
groups, cur = [], []
for r in rows:
    if cur and r[1] - max(x[2] for x in cur) > 1_000_000:
        groups.append(cur); cur = []
    cur.append(r)
groups.append(cur)
g = groups[-1]
# proofs run back to back: keep the last one = from the last kernel whose name contains the marker (default: first kernel name of a proof)
marker = sys.argv[2] if len(sys.argv) > 2 else "k_eval_witness"
idx = [i for i, r in enumerate(g) if marker in r[0]]
if idx:
    g = g[idx[-1]:]
t0 = g[0][1]
busy_end = t0
short = lambda n: re.sub(r"\(.*", "", n.replace("plk::", "").replace("void ", ""))[:34]
print("last group: %d kernels, %.3f ms from first start to last end" % (len(g), (max(x[2] for x in g) - t0) / 1e6))
idle = 0
for name, s, e, st, gx, wx in g:
    gap = s - busy_end
    if gap > 0:
        idle += gap
    print("%9.3f  +%7.3f ms  st%-3d gap %7.1f us  %-34s grid %d" % ((s - t0) / 1e6, (e - s) / 1e6, st, max(gap, 0) / 1e3, short(name), gx // max(wx, 1)))
    busy_end = max(busy_end, e)
print("GPU idle inside the group: %.3f ms" % (idle / 1e6))
