#!/bin/bash
# idle time between the kernels of the tile line's TIMED window (8 tiles on one GPU: 480 tile-ticks ahead of the 480 replayed ones)
SC=${1:-config4}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace -d $O/gt_${SC}_trace -o bench -- python $R/bench.py --gpus 1 --scene $SC --no-cpu-baseline > $O/gt_${SC}.log 2>&1
( cd $R; python tools/scratch/gaps.py gpurun_out/gt_${SC}_trace/bench_results.db 480 480 | head -${HEAD:-40}; rm -rf gpurun_out/gt_${SC}_trace )
