#!/bin/bash
# The round's rocprofv3 evidence (run on the GPU box through gpurun): for every window bench.py reports - the falling pile (default:
# --warmup 10 --steps 60), the driver's own command (--warmup 5 --steps 20), the settled pile (--warmup 400), BASELINE configs 3 and 5,
# each with 60 and with 20 timed ticks (the driver's nested windows) - a kernel trace and the two PMC passes (FETCH_SIZE, WRITE_SIZE:
# separate runs, counters only - MI355X_MICROARCH.md), all on the exact bench command.
#   Usage: bash tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>_*      then: python tools/publish_profiles.py <tag>
#   (WINDOWS="driver transient" NO_TILES=1 ONLY_BENCH_LINES= ... : a subset of the windows; the final bench lines are always taken)
set -u
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
B="python $R/bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs --min-seconds 0"
cd /tmp && export TMPDIR=/tmp
run() {  # name, timed ticks, dominant kernel, window label for bench.py, its warm-up, bench args...
  local N=$1 K=$2 KER=$3 WIN=$4 W=$5; shift 5
  if [ -n "${WINDOWS:-}" ] && ! echo " $WINDOWS " | grep -q " $N "; then return; fi  # (WINDOWS="driver transient": only those)
  rocprofv3 --kernel-trace -d $O/${TAG}_${N}_trace -o bench -- $B "$@" > $O/${TAG}_${N}_trace.log 2>&1
  grep -a '"metric"' $O/${TAG}_${N}_trace.log | tail -1 > $O/${TAG}_${N}_bench_under_trace.json
  rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_${N}_fetch -o bench -- $B "$@" > $O/${TAG}_${N}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_${N}_write -o bench -- $B "$@" > $O/${TAG}_${N}_write.log 2>&1
  ( cd $R
    python tools/rocprof_summary.py gpurun_out/${TAG}_${N}_trace/bench_results.db $K --timed $KER $K > gpurun_out/${TAG}_${N}_kernel_stats.txt
    python tools/pmc_summary.py gpurun_out/${TAG}_${N}_fetch/bench_results.db gpurun_out/${TAG}_${N}_write/bench_results.db --timed $KER $K gpurun_out/${TAG}_${N}_pmc.json $WIN $W $K > gpurun_out/${TAG}_${N}_pmc_hbm_traffic.txt
    rm -rf gpurun_out/${TAG}_${N}_trace gpurun_out/${TAG}_${N}_fetch gpurun_out/${TAG}_${N}_write
    python tools/publish_profiles.py $TAG > /dev/null )  # (the box's copy of profiles/: the run below reports this pass's traffic)
  $B "$@" > $O/${TAG}_${N}_bench.json 2> /dev/null
}
if [ -z "${ONLY_TILES:-}" ]; then
run transient  60 k_solve_flow6 transient 10 --warmup 10 --steps 60
run driver     20 k_solve_flow6 transient 5  --warmup 5 --steps 20
run settled    60 k_solve_flow6 settled 400 --warmup 400 --steps 60
run settled20  20 k_solve_flow6 settled 400 --warmup 400 --steps 20
run config3    60 k_solve_flow6 config3 150 --scene config3
run config3_20 20 k_solve_flow6 config3 150 --scene config3 --steps 20
run config5    60 k_solve_flow6 config5 80  --scene config5
run config5_20 20 k_solve_flow6 config5 80  --scene config5 --steps 20
fi
# the tile launches (4 + 4 + 2 iterations of one 131 072-body tile, a ghost velocity refresh between them - R = 4; 8 tiles on this GPU): the
# instrumented ticks' 60 x 8 x 3 launches are the last 1440 of the run (TILE_LAUNCHES: 2400 at R = 2)
run_tiles() {  # name, scene, warm-up
  local N=$1 SC=$2 W=$3
  local BT="python $R/bench.py --gpus 1 --scene $SC --no-cpu-baseline --no-settled-tiles"  # (the profiled launches are the LAST ones of the run: the replay of the timed falling-pile ticks, not the settled window behind it)
  rocprofv3 --kernel-trace -d $O/${TAG}_${N}_trace -o bench -- $BT > $O/${TAG}_${N}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_${N}_fetch -o bench -- $BT > $O/${TAG}_${N}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_${N}_write -o bench -- $BT > $O/${TAG}_${N}_write.log 2>&1
  ( cd $R
    python tools/rocprof_summary.py gpurun_out/${TAG}_${N}_trace/bench_results.db 480 --tick-start k_integrate --timed k_solve_flow6 ${TILE_LAUNCHES:-1440} > gpurun_out/${TAG}_${N}_kernel_stats.txt
    python tools/pmc_summary.py gpurun_out/${TAG}_${N}_fetch/bench_results.db gpurun_out/${TAG}_${N}_write/bench_results.db --timed k_solve_flow6 ${TILE_LAUNCHES:-1440} gpurun_out/${TAG}_${N}_pmc.json tiles $W 60 > gpurun_out/${TAG}_${N}_pmc_hbm_traffic.txt
    rm -rf gpurun_out/${TAG}_${N}_trace gpurun_out/${TAG}_${N}_fetch gpurun_out/${TAG}_${N}_write
    python tools/publish_profiles.py $TAG > /dev/null )
}
if [ -z "${NO_TILES:-}" ]; then
run_tiles config4_tiles config4 10
run_tiles config5_tiles config5_tiles 80
fi
cd $R
if [ -z "${ONLY_TILES:-}" ]; then
python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_command.json 2> /dev/null
python bench.py --scene config5 --no-cpu-baseline > $O/${TAG}_config5_undivided_1gpu.json 2> /dev/null
python tools/config4_undivided.py > /dev/null 2>&1; cp $O/config4_undivided_1gpu.json $O/${TAG}_config4_undivided_1gpu.json
fi
python bench.py --gpus 1 --scene config4 --no-cpu-baseline > $O/${TAG}_config4_8tiles_1gpu_bench.json 2> /dev/null
python bench.py --gpus 1 --scene config5_tiles --no-cpu-baseline > $O/${TAG}_config5_8tiles_1gpu_bench.json 2> /dev/null
for RR in 2 3 4 5; do python bench.py --gpus 1 --scene config4 --no-cpu-baseline --refresh-every $RR > $O/${TAG}_config4_tiles_refresh_every_$RR.json 2> /dev/null; done  # (the seam against R, the pile at rest)
ls $O | grep ${TAG}_
