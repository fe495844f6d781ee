#!/bin/bash
# Issue / wait counters of a bench window's kernels (one --pmc pass per group; counters only).
#   BENCH_ARGS="--scene config3 --steps 20" KPAT="pair_grid|terrain" bash tools/r06/pmc_kernels.sh "<counters of pass 1>" "<counters of pass 2>" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for G in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $G -d $O/pmck_$i -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs --min-seconds 0 ${BENCH_ARGS:---steps 20 --warmup 5} > $O/pmck_$i.log 2>&1
  KPAT="${KPAT:-.}" python - <<PY
import sqlite3, glob, collections, os, re
db = sqlite3.connect(glob.glob("$O/pmck_$i/*results.db")[0])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for k, c, v in db.execute("select kernel_name, counter_name, value from counters_collection order by start"):
    acc[k.split("(")[0][-40:]][c].append(float(v))
names = sorted({c for k in acc for c in acc[k]})
print("%-42s" % "kernel (last 20 launches, mean)", " ".join("%18s" % c[:18] for c in names))
for k in acc:
    if re.search(os.environ["KPAT"], k):
        print("%-42s" % k, " ".join("%18.0f" % (sum(acc[k][c][-20:]) / max(len(acc[k][c][-20:]), 1)) for c in names))
PY
  rm -rf $O/pmck_$i
done
