"""idle time between consecutive kernels of the last K ticks of a rocprofv3 kernel trace (rocpd sqlite)"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1]); K = int(sys.argv[2]); SKIP = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # SKIP: ticks behind the window (bench.py's instrumented replays)
rows = db.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "k_tick_clear" in r[0] or "k_reset_step" in r[0]]
first = starts[-(K + SKIP)]
last = starts[-SKIP] if SKIP else len(rows)
gap, cnt = defaultdict(float), defaultdict(int)
tot = 0.0
for i in range(first + 1, last):
    a, b = rows[i - 1], rows[i]
    g = (b[1] - a[2]) / 1e3
    key = a[0].split("(")[0].replace("void mgf::", "").replace("mgf::", "")[:28] + " -> " + b[0].split("(")[0].replace("void mgf::", "").replace("mgf::", "")[:28]
    gap[key] += g; cnt[key] += 1; tot += g
print(f"idle between kernels: {tot / K:.1f} us per tick over the last {K} ticks")
for k, v in sorted(gap.items(), key=lambda kv: -kv[1]):
    print(f"  {k:60s} {v / cnt[k]:7.2f} us x {cnt[k]}")
