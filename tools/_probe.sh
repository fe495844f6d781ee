cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_tiles_native.py tests/test_gpu_migration.py tests/test_gpu_edge_cases.py tests/test_compound_bodies.py -m gpu -x -q 2>&1 | grep -a "passed\|failed" | tail -3
python tools/tile_native_timing.py 2>&1 | tail -1
for k in 1 2; do
python bench.py --gpus 1 --scene config4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4 on 1 GPU:', d['ms_per_step'], d['value'])"
done
