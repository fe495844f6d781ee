#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md prescribes).  Counter unit: KiB.  On gfx950 FETCH_SIZE reads exactly 1/2 of a wide
coalesced read stream (guide §HBM); both the raw and the x2-corrected read side are reported.

    python tools/pmc_summary.py gpurun_out/pmc_fetch/pmc_counter_collection.csv gpurun_out/pmc_write/pmc_counter_collection.csv [out.json]
    (or the two pmc_results.db files of rocprofv3's default rocpd output)
    ... --timed <kernel substring> <count> <out.json>: the average over the LAST <count> launches of the kernels whose
    name holds the substring (bench.py's timed region; .db inputs only), written in the form bench.py reads.
"""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    if path.endswith(".db"):  # rocprofv3's rocpd sqlite output (view counters_collection)
        import sqlite3
        db = sqlite3.connect(path)
        for name, value in db.execute("select kernel_name, value from counters_collection where counter_name = ? order by start", (counter,)):
            name = name.split("(")[0]
            tot[name] += float(value)
            cnt[name] += 1
        return tot, cnt
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            tot[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return tot, cnt


def last_launches(path, counter, pattern, count):
    import sqlite3
    db = sqlite3.connect(path)
    vals = [float(v) for (v,) in db.execute("select value from counters_collection where counter_name = ? and kernel_name like ? order by start",
                                            (counter, f"%{pattern}%"))]
    vals = vals[-count:]
    return sum(vals) * 1024.0 / len(vals), len(vals)


def lib_sha256():
    """sha256 of the libmgf_hip.so the passes ran (MGF_AMD_LIB, or the tree's): bench.py reports a record's traffic only while it is
    running that very library - a kernel change without fresh passes must not keep printing the old bytes"""
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.environ.get("MGF_AMD_LIB") or os.path.join(root, "mgf_amd", "libmgf_hip.so")
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()
    except OSError:
        return None


def main():
    if "--timed" in sys.argv:
        k = sys.argv.index("--timed")
        pattern, count, out = sys.argv[k + 1], int(sys.argv[k + 2]), sys.argv[k + 3]
        # optional: the bench window the passes were taken on (name, warm-up ticks, timed ticks) - bench.py reports the traffic only
        # beside a run of the same window
        window = [sys.argv[k + 4], int(sys.argv[k + 5]), int(sys.argv[k + 6])] if len(sys.argv) > k + 6 else None
        f, n = last_launches(sys.argv[1], "FETCH_SIZE", pattern, count)
        w, _ = last_launches(sys.argv[2], "WRITE_SIZE", pattern, count)
        d = dict(kernel=pattern, launches=n, fetch_bytes_per_launch_raw=f, write_bytes_per_launch=w, hbm_bytes_per_launch_raw=f + w,
                 hbm_bytes_per_launch=2 * f + w, lib_sha256=lib_sha256())
        if window:
            d["window"] = window
            d["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --no-cpu-baseline --no-settled "
                         f"--no-order-check --min-seconds 0 --warmup {window[1]}` (262144 spheres, {window[2]} timed ticks, tools/collect_profiles.sh); "
                         f"average over the LAST {count} launches of {pattern} = the timed window; counter unit KiB; hbm_bytes_per_launch = 2*FETCH + WRITE "
                         "(gfx950 FETCH_SIZE reads 1/2 of 16-B/lane loads, MI355X_MICROARCH.md HBM section)")
        json.dump(d, open(out, "w"), indent=1)
        print(f"# {pattern}: last {n} launches: fetch {f:.0f} B, write {w:.0f} B, raw {f + w:.0f} B, fetch x2 {2 * f + w:.0f} B per launch")
        sys.argv = sys.argv[:k]
    fetch, fc = load(sys.argv[1], "FETCH_SIZE")
    write, wc = load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in fetch:
        n = fc[k]
        f = fetch[k] * 1024.0 / n
        w = write.get(k, 0.0) * 1024.0 / max(wc.get(k, 1), 1)
        rows.append(dict(kernel=k, launches=n, fetch_bytes_per_launch_raw=f, write_bytes_per_launch=w,
                         hbm_bytes_per_launch_raw=f + w, hbm_bytes_per_launch_fetch_x2=2 * f + w))
    rows.sort(key=lambda r: -r["hbm_bytes_per_launch_raw"] * r["launches"])
    print(f"{'kernel':60s} {'launches':>8s} {'fetch/launch':>14s} {'write/launch':>14s} {'total raw':>14s} {'total fetchx2':>14s}")
    for r in rows[:25]:
        print(f"{r['kernel'][:60]:60s} {r['launches']:8d} {r['fetch_bytes_per_launch_raw']:14.0f} {r['write_bytes_per_launch']:14.0f} "
              f"{r['hbm_bytes_per_launch_raw']:14.0f} {r['hbm_bytes_per_launch_fetch_x2']:14.0f}")
    if len(sys.argv) > 3:
        json.dump(rows, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
