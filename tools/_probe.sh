cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_compound_bodies.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | grep -a "passed\|failed" | tail -3
B="python bench.py --no-cpu-baseline --no-order-check --no-settled --min-seconds 1"
for k in 1 2; do
$B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['value']/1e9,3), d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done
