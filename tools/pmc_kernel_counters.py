#!/usr/bin/env python
"""Per-kernel averages of arbitrary rocprofv3 --pmc counters (rocpd .db outputs), e.g. the SQ instruction / stall counters:
    python tools/pmc_kernel_counters.py <kernel substring> a_results.db [b_results.db ...]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    pat = sys.argv[1]
    for path in sys.argv[2:]:
        db = sqlite3.connect(path)
        tot, cnt = defaultdict(float), defaultdict(int)
        for name, counter, value in db.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like ?", (f"%{pat}%",)):
            key = (name.split("(")[0][-40:], counter)
            tot[key] += float(value)
            cnt[key] += 1
        for (name, counter), v in sorted(tot.items()):
            print(f"{name:42s} {counter:28s} {v / cnt[(name, counter)]:16.1f}   ({cnt[(name, counter)]} launches)")


if __name__ == "__main__":
    main()
