#!/bin/bash
# A/B of a library option on one scene: kernel-trace summaries of the bench command's 20 timed ticks.
#   bash tools/scratch/ab_kernels.sh <scene> <opt=a> <opt=b>
SC=$1; A=$2; Bv=$3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in $A $Bv; do
  T=$O/ab_${SC}_${V//=/_}
  rocprofv3 --kernel-trace -d ${T}_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs --min-seconds 0 --scene $SC --steps 20 --opt $V > ${T}.log 2>&1
  ( cd $R; python tools/rocprof_summary.py gpurun_out/ab_${SC}_${V//=/_}_trace/bench_results.db 20 --timed k_solve_flow6 20 > gpurun_out/ab_${SC}_${V//=/_}_kernel_stats.txt; rm -rf gpurun_out/ab_${SC}_${V//=/_}_trace )
  echo "== $SC $V"; cut -c1-60,75-140 $O/ab_${SC}_${V//=/_}_kernel_stats.txt | head -24
done
