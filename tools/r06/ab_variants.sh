#!/bin/bash
# same-box kernel traces of the tree's library and of variant builds: bash tools/r06/ab_variants.sh "<variant names>" <timed ticks> <bench args...>
VS=$1; K=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for L in tree $VS; do
  if [ $L = tree ]; then unset MGF_AMD_LIB; else export MGF_AMD_LIB=$R/mgf_amd/variants/libmgf_hip_$L.so; fi
  rocprofv3 --kernel-trace -d $O/abv_${L}_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs --min-seconds 0 "$@" > $O/abv_${L}.log 2>&1
  ( cd $R; python tools/rocprof_summary.py gpurun_out/abv_${L}_trace/bench_results.db $K --timed k_solve_flow6 $K > gpurun_out/abv_${L}_kernel_stats.txt; rm -rf gpurun_out/abv_${L}_trace )
  echo "== $L"; cut -c1-60,75-140 $O/abv_${L}_kernel_stats.txt | head -${HEAD:-9}
done
