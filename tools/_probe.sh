cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tiles_native.py tests/test_compound_bodies.py tests/test_world_snapshots.py -m gpu -x -q 2>&1 | grep -a "passed\|failed" | tail -3
B="python bench.py --no-cpu-baseline --no-settled --no-order-check --min-seconds 1"
for k in 1 2; do
$B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['value']/1e9,3), d['roofline']['avg_launch_us'], d['phase_ms_per_step_rank0'])"
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/pk_trace -o bench -- python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --min-seconds 0 > $R/gpurun_out/pk_trace.log 2>&1
cd $R
python tools/rocprof_summary.py gpurun_out/pk_trace/bench_results.db 60 | head -14 | cut -c1-150
