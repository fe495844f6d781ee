set -u
cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_gpu_multi_device.py tests/test_gpu_tiles_native.py tests/test_gpu_migration.py tests/test_compound_bodies.py -x -q -m gpu 2>&1 | tail -8)
(time python bench.py --min-seconds 0.3 > gpurun_out/r03e_bench_full.json 2> gpurun_out/r03e_bench_full.err) 2>&1 | tail -4
tail -3 gpurun_out/r03e_bench_full.err
python bench.py --scene config3 --no-cpu-baseline --min-seconds 0.3 > gpurun_out/r03e_bench_config3.json 2>/dev/null
python bench.py --scene config5 --no-cpu-baseline --min-seconds 0.3 > gpurun_out/r03e_bench_config5.json 2>/dev/null
python bench.py --gpus 1 --scene config4 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03e_bench_config4_8tiles.json 2>/dev/null
