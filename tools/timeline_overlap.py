#!/usr/bin/env python3
"""two proofs in flight: from a rocprofv3 kernel trace (rocpd database) of tools/prove_inflight_probe.py, a window of the
concurrent phase — every kernel with the prover it belongs to (the host thread that launched it: one thread per proof in
flight), start, duration, and what the OTHER prover was running at that moment.  Shows msm_accumulate of one proof under the reduction tails / point-wise kernels of the other.
usage: tools/timeline_overlap.py <results.db> [window_ms=22]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 22.0
rows = db.execute("select name, start, end, tid from kernels order by start").fetchall()
short = lambda n: re.sub(r"[<(].*", "", n.replace("plk::", "").replace("void ", ""))[:26]
# the concurrent phase is the tail of the trace: take the last `win` ms ending 5 ms before the last kernel
t_end = rows[-1][2] - 5_000_000
t_beg = t_end - int(win * 1e6)
sel = [r for r in rows if r[1] >= t_beg and r[1] < t_end]
sel = [r for r in sel if r[2] <= t_end + 2_000_000]
tids = sorted({r[3] for r in sel})
owner = {t: i for i, t in enumerate(tids)}
print("window %.1f ms, %d kernels, launching host threads %s -> provers %s" % (win, len(sel), tids, "".join("ABCD"[owner[t]] for t in tids)))
if len(tids) != 2:
    sys.exit("expected exactly two launching threads in the window (two proofs in flight)")
acc = [r for r in sel if "msm_accumulate" in r[0]]
def running(other, t):
    return [short(r[0]) for r in sel if owner[r[3]] == other and r[1] <= t < r[2]]
print("%9s %8s  %-2s %-26s | the other prover at that moment" % ("start ms", "dur ms", "", "kernel"))
for name, s, e, st in sel:
    if e - s < 20_000 and "msm_accumulate" not in name:
        continue                                                       # (kernels under 20 us are left out of the listing)
    o = owner[st]
    mid = (s + e) // 2
    print("%9.3f %8.3f  %s  %-26s | %s" % ((s - t_beg) / 1e6, (e - s) / 1e6, "AB"[o], short(name), ", ".join(sorted(set(running(1 - o, mid)))) or "-"))
# how much of every accumulation ran beside kernels of the other prover
tot = ov = 0
for name, s, e, st in acc:
    o = owner[st]
    tot += e - s
    iv = sorted((max(s, r[1]), min(e, r[2])) for r in sel if owner[r[3]] != o and r[2] > s and r[1] < e)
    cur = s
    for a, b in iv:
        a = max(a, cur)
        if b > a:
            ov += b - a; cur = b
busy = 0
cur = t_beg
for name, s, e, st in sorted(sel, key=lambda r: r[1]):
    a = max(s, cur)
    e = min(e, t_end)
    if e > a:
        busy += e - a; cur = e
print("msm_accumulate: %d launches, %.2f ms in total, %.1f %% of that time with a kernel of the OTHER prover also running" % (len(acc), tot / 1e6, 100.0 * ov / max(tot, 1)))
print("GPU busy (any kernel) %.1f %% of the window" % (100.0 * busy / (t_end - t_beg)))
