#!/bin/bash
# Same-box A/B of two builds of libmgf_hip.so on the driver's window: kernel-trace summaries.
#   bash tools/scratch/ab_lib.sh <variant .so (relative to the repo)> [bench args...]
V=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for L in tree variant; do
  if [ $L = variant ]; then export MGF_AMD_LIB=$R/$V; else unset MGF_AMD_LIB; fi
  rocprofv3 --kernel-trace -d $O/abl_${L}_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs --min-seconds 0 --steps 20 --warmup 5 "$@" > $O/abl_${L}.log 2>&1
  ( cd $R; python tools/rocprof_summary.py gpurun_out/abl_${L}_trace/bench_results.db 20 --timed k_solve_flow6 20 > gpurun_out/abl_${L}_kernel_stats.txt; rm -rf gpurun_out/abl_${L}_trace )
  echo "== $L"; cut -c1-60,75-140 $O/abl_${L}_kernel_stats.txt | head -12
done
