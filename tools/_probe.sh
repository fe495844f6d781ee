cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -a "passed\|failed" | tail -3
B="python bench.py --no-cpu-baseline --no-order-check --min-seconds 1"
for k in 1 2; do
$B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['value']/1e9,3), d['roofline']['avg_launch_us'], d['roofline']['frac'], d['settled']['ms_per_step'], d['settled']['roofline']['avg_launch_us'], d['settled']['roofline']['frac'])"
done
