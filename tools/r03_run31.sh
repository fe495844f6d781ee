set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_multi_device.py tests/test_gpu_tiles_native.py tests/test_gpu_migration.py -m gpu -x -q > gpurun_out/r03_pytest31.log 2>&1; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r03_pytest31.log | tail -3
python bench.py --gpus 1 --scene config4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config4 8 tiles ms/tick', round(d['ms_per_step'],3), 'tile tick', round(d['tile_tick_ms_rank0'],4), 'G/s', round(d['value']/1e9,3), 'frac', d['roofline']['frac'], d['exchange']['per_rank'][0])"
export MGF_RCCL_LIB=$GRAFT_REPO_ROOT/tests/fake_rccl/libmgf_fake_rccl.so MGF_BENCH_DEVICE=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 10 --warmup 3 --backend gloo 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('2 ranks/1 gpu shim: ms/tick', round(d['ms_per_step'],3), d['exchange']['per_rank'], d['seam_penetration']['across_tile_faces'])"
