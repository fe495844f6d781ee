set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/r03q_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs --min-seconds 0 > $R/gpurun_out/r03q_trace.log 2>&1
cd $R; python tools/trace_gaps.py gpurun_out/r03q_trace/bench_results.db 60; rm -rf gpurun_out/r03q_trace
