#!/bin/bash
# A/B of library options on one bench window: kernel-trace summaries of the bench command's timed ticks.
#   bash tools/scratch/ab_opt.sh "<bench args, e.g. --steps 20 --warmup 5>" <ticks> <opt=a> <opt=b> ...
ARGS=$1; K=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  T=$O/abo_${V//=/_}
  rocprofv3 --kernel-trace -d ${T}_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs --min-seconds 0 $ARGS --opt $V > ${T}.log 2>&1
  ( cd $R; python tools/rocprof_summary.py gpurun_out/abo_${V//=/_}_trace/bench_results.db $K --timed k_solve_flow6 $K > gpurun_out/abo_${V//=/_}_kernel_stats.txt; rm -rf gpurun_out/abo_${V//=/_}_trace )
  echo "== $ARGS $V"; cut -c1-60,75-140 $O/abo_${V//=/_}_kernel_stats.txt | head -24
done
