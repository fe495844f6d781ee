set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/r03s_trace -o bench -- python $R/tools/plain_ticks.py 10 60 > $R/gpurun_out/r03s_trace.log 2>&1
cd $R; grep "ms/tick" gpurun_out/r03s_trace.log; python tools/trace_gaps.py gpurun_out/r03s_trace/bench_results.db 60 | head -14; rm -rf gpurun_out/r03s_trace
python tools/plain_ticks.py 10 60
