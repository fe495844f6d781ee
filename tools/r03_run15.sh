set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo ---- bench 2 ranks on one GPU over the stand-in transport
export MGF_RCCL_LIB=$GRAFT_REPO_ROOT/tests/fake_rccl/libmgf_fake_rccl.so MGF_BENCH_DEVICE=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --backend gloo 2>&1 | tail -3
unset MGF_RCCL_LIB MGF_BENCH_DEVICE
echo ---- bench config4 8 tiles on 1 gpu
timeout 900 python bench.py --gpus 1 --scene config4 --no-cpu-baseline 2>&1 | tail -1
