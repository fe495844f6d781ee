set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for S in config2 config3 config5; do
rocprofv3 --kernel-trace -d $R/gpurun_out/r03j_${S}_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs --min-seconds 0 --scene $S > $R/gpurun_out/r03j_${S}_trace.log 2>&1
( cd $R; python tools/rocprof_summary.py gpurun_out/r03j_${S}_trace/bench_results.db 60 --timed k_solve_flow6 60 > gpurun_out/r03j_${S}_kernel_stats.txt; rm -rf gpurun_out/r03j_${S}_trace )
done
