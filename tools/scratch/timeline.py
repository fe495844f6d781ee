import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); SKIP = int(sys.argv[2])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "k_tick_clear" in r[0] or "k_reset_step" in r[0]]
i0, i1 = starts[-(SKIP + 1)], starts[-SKIP]
t0 = rows[i0][1]
for r in rows[i0:i1 + 1]:
    print(f"{(r[1] - t0) / 1e3:9.2f} +{(r[2] - r[1]) / 1e3:8.2f}  {r[0].split('(')[0].replace('void mgf::', '').replace('mgf::', '')[:50]}")
