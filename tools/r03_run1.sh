set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r03a_pytest.log
B="python bench.py --no-cpu-baseline --no-order-check --min-seconds 0.5"
for R in 0 4 16 64; do
  $B --opt resort_every=$R > gpurun_out/r03a_bench_resort$R.json 2> gpurun_out/r03a_bench_resort$R.err
done
cd /tmp && export TMPDIR=/tmp
for R in 0 16; do
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r03a_trace$R -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-settled --no-order-check --min-seconds 0 --opt resort_every=$R > $GRAFT_REPO_ROOT/gpurun_out/r03a_trace$R.log 2>&1
done
cd $GRAFT_REPO_ROOT
for R in 0 16; do python tools/rocprof_summary.py gpurun_out/r03a_trace$R/bench_results.db 60 > gpurun_out/r03a_kernel_stats_resort$R.txt; done
rm -rf gpurun_out/r03a_trace0 gpurun_out/r03a_trace16
cat gpurun_out/r03a_pytest.log
