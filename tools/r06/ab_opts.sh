#!/bin/bash
# same-box kernel traces of one bench window under different library options: bash tools/r06/ab_opts.sh "<opt=a,opt2=b> <...>" <timed ticks> <bench args...>
# ("none" = no option; several options of one run are joined by commas)
VS=$1; K=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in $VS; do
  OPTS=""; if [ $V != none ]; then for o in ${V//,/ }; do OPTS="$OPTS --opt $o"; done; fi
  T=abo_${V//[=,]/_}
  rocprofv3 --kernel-trace -d $O/${T}_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs --min-seconds 0 "$@" $OPTS > $O/${T}.log 2>&1
  ( cd $R; python tools/rocprof_summary.py gpurun_out/${T}_trace/bench_results.db $K --timed k_solve_flow6 $K > gpurun_out/${T}_kernel_stats.txt; rm -rf gpurun_out/${T}_trace )
  echo "== $V"; cut -c1-60,75-140 $O/${T}_kernel_stats.txt | head -${HEAD:-9}
done
