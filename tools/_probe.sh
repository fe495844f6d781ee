cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/c4_trace -o bench -- python $R/bench.py --gpus 1 --scene config4 --no-cpu-baseline --steps 20 --warmup 5 > $R/gpurun_out/c4_trace.log 2>&1
cd $R
python tools/rocprof_summary.py gpurun_out/c4_trace/bench_results.db 160 | head -24 | cut -c1-150
