import sqlite3, sys, glob
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0])
cols = [r[1] for r in db.execute("pragma table_info(regions)").fetchall()]
print(cols)
rows = db.execute("select name, start, end from regions order by start").fetchall()
t0 = rows[0][1]
big = [(n, (s - t0) / 1e6, (e - s) / 1e6) for n, s, e in rows if e - s > 500_000]
print("calls longer than 0.5 ms (name, at ms, took ms):")
for n, at, d in big: print("  %-32s %9.1f %8.2f" % (n[:32], at, d))
agg = {}
for n, s, e in rows: a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e6
print("by name (top 15 by total ms):")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:15]: print("  %-32s %6d calls %9.2f ms" % (n[:32], c, t))
