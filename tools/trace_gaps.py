"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace database, by the pair of kernels around the gap (development aid)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if 'k_tick_clear' in r[0]]
first = marks[-steps] if len(marks) >= steps else 0
gaps = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
short = lambda s: s.split('(')[0].replace('void mgf::', '').replace('mgf::', '')[:28]
for i in range(first, len(rows) - 1):
    g = (rows[i + 1][1] - rows[i][2]) / 1e3
    k = (short(rows[i][0]), short(rows[i + 1][0]))
    gaps[k][0] += 1; gaps[k][1] += g; tot += g
print(f"total gap time {tot / 1e3:.3f} ms over the last {steps} ticks = {tot / steps:.1f} us per tick")
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:20]:
    print(f"{k[0]:30s} -> {k[1]:30s} {c:5d} x {t / c:7.2f} us")
