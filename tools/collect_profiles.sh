#!/bin/bash
# rocprofv3 evidence for bench.py's default window and for the settled pile: kernel trace, then the two PMC passes (separate
# runs, counters only - MI355X_MICROARCH.md).  Usage on the GPU box: bash tools/collect_profiles.sh <tag>   (writes gpurun_out/<tag>_*)
set -u
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --min-seconds 0"
cd /tmp && export TMPDIR=/tmp
for W in 10 400; do
  N=t; [ $W = 400 ] && N=s
  rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}${N}_trace -o bench -- $B --warmup $W > $R/gpurun_out/${TAG}${N}_trace.log 2>&1
  grep -a '"metric"' $R/gpurun_out/${TAG}${N}_trace.log | tail -1 > $R/gpurun_out/${TAG}${N}_bench_under_trace.json
  rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}${N}_fetch -o bench -- $B --warmup $W > $R/gpurun_out/${TAG}${N}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}${N}_write -o bench -- $B --warmup $W > $R/gpurun_out/${TAG}${N}_write.log 2>&1
  $B --warmup $W > $R/gpurun_out/${TAG}${N}_bench.json 2> /dev/null
done
cd $R
for N in t s; do
  python tools/rocprof_summary.py gpurun_out/${TAG}${N}_trace/bench_results.db 60 > gpurun_out/${TAG}${N}_kernel_stats.txt
  W=10; NAME=transient; [ $N = s ] && W=400 && NAME=settled
  python tools/pmc_summary.py gpurun_out/${TAG}${N}_fetch/bench_results.db gpurun_out/${TAG}${N}_write/bench_results.db --timed k_solve_flow6 60 gpurun_out/${TAG}${N}_pmc_k_solve_flow6.json $NAME $W 60 > gpurun_out/${TAG}${N}_pmc_hbm_traffic.txt
done
ls gpurun_out | grep ${TAG}
