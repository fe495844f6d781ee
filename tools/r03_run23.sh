set -u
cd $GRAFT_REPO_ROOT
( time python bench.py > gpurun_out/r03l_bench_default.json 2> gpurun_out/r03l_bench_default.err ) 2>&1 | tail -3
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r03l_bench_driver.json 2> /dev/null ) 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
