set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2000 python -m pytest tests -q -m gpu 2>&1 | tail -15) > gpurun_out/r03b_pytest.log
B="python bench.py --no-cpu-baseline --no-order-check --min-seconds 0.5"
for R in 0 8 32; do
  $B --opt resort_every=$R > gpurun_out/r03b_bench_resort$R.json 2> gpurun_out/r03b_bench_resort$R.err
done
cat gpurun_out/r03b_pytest.log
