"""one tick of a rocprofv3 kernel trace as a timeline: start and end of every launch relative to the tick's first one (us), and the queue it ran on"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select name, start, end{', ' + q if q else ''} from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "k_tick_clear" in r[0]]
a, b = starts[-back], starts[-back + 1] if back > 1 else len(rows)
t0 = rows[a][1]
print("columns:", cols)
for r in rows[a:b]:
    print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - t0) / 1e3:8.1f}  {(r[2] - r[1]) / 1e3:7.1f}  q={r[3] if q else '-'}  {r[0].split('(')[0].replace('void mgf::', '').replace('mgf::', '')[:60]}")
print(f"tick: {(rows[b - 1][2] - t0) / 1e3:.1f} us from the first launch's start to the last one's end; next tick starts {(rows[b][1] - rows[b - 1][2]) / 1e3 if b < len(rows) else 0:.1f} us later")
