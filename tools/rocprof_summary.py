#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace database (rocpd sqlite) per kernel.

    python tools/rocprof_summary.py gpurun_out/prof_x/bench_results.db [steps] > profiles/rNN_kernel_stats.txt
"""
import json
import sqlite3
import sys


def window_start(path, steps, tick_start=None):
    """start time of the `steps`-th tick from the end (a tick begins with its clearing launch: k_tick_clear, k_reset_step before round
    3's merge; `tick_start`: another kernel that runs once per tick - the tile set resets all its tiles in one launch, so its tile-ticks
    are counted by their k_integrate): bench.py's timed region - or None when the trace holds fewer ticks (then everything is summarised)"""
    db = sqlite3.connect(path)
    if tick_start:
        rows = db.execute("select start from kernels where name like ? order by start desc limit ?", (f"%{tick_start}%", steps)).fetchall()
    else:
        rows = db.execute("select start from kernels where name like '%k_tick_clear%' or name like '%k_reset_step%' order by start desc limit ?", (steps,)).fetchall()
    return rows[-1][0] if len(rows) == steps else None


def summarise(path, t0=None):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                      "max(vgpr_count), max(sgpr_count), max(scratch_size), max(lds_size) from kernels where start >= ? group by name "
                      "order by 3 desc", (t0 or 0,)).fetchall()
    return [dict(name=r[0], calls=r[1], total_us=r[2] / 1e3, avg_us=r[3] / 1e3, min_us=r[4] / 1e3, max_us=r[5] / 1e3,
                 vgpr=r[6], sgpr=r[7], scratch=r[8], lds=r[9]) for r in rows]


def last_launches(path, pattern, count):
    """average duration (us) of the last `count` launches of the kernels matching `pattern` (bench.py's timed region)"""
    db = sqlite3.connect(path)
    rows = db.execute("select end-start from kernels where name like ? order by start desc limit ?", (f"%{pattern}%", count)).fetchall()
    return sum(r[0] for r in rows) / 1e3 / max(len(rows), 1), len(rows)


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else None
    tick_start = sys.argv[sys.argv.index("--tick-start") + 1] if "--tick-start" in sys.argv else None
    t0 = window_start(path, steps, tick_start) if steps else None
    rows = summarise(path, t0)
    tot = sum(r["total_us"] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of {path}" + (f": the last {steps} ticks (bench.py's timed region; warm-up launches left out)" if t0 else ""))
    print(f"# total kernel time {tot / 1e3:.3f} ms" + (f" over {steps} ticks = {tot / 1e3 / steps:.3f} ms/tick" if steps and t0 else ""))
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'%':>6s} {'vgpr':>5s} {'scratch':>7s}")
    for r in rows:
        print(f"{r['name'][:70]:70s} {r['calls']:7d} {r['total_us'] / 1e3:10.3f} {r['avg_us']:9.2f} {r['min_us']:8.2f} "
              f"{r['max_us']:9.2f} {100 * r['total_us'] / tot:6.1f} {r['vgpr'] or 0:5d} {r['scratch'] or 0:7d}")
    if "--timed" in sys.argv:  # --timed <kernel substring> <launches in bench.py's timed region>
        k = sys.argv.index("--timed")
        avg, cnt = last_launches(path, sys.argv[k + 1], int(sys.argv[k + 2]))
        print(f"# {sys.argv[k + 1]}: average of the last {cnt} launches (bench.py's timed region) = {avg:.2f} us")
    if "--json" in sys.argv:
        print(json.dumps(rows))


if __name__ == "__main__":
    main()
