#!/bin/bash
# Calibration for a per-block choice of "records in LDS" (RL): what RL is worth on a tile's 2-iteration launches and on config 5.
cd $GRAFT_REPO_ROOT
O=gpurun_out/rl_calib.jsonl
rm -f $O
S="python tools/r04_block_sweep.py --nb 0 --out $O"
$S --scenes tile,config5,config3
$S --scenes tile --opt flow6_rec_lds=2 --opt flow6_slot_margin=32
$S --scenes config5 --opt flow6_rec_lds=0
$S --scenes config5 --opt flow6_rec_lds=2 --opt flow6_const_lds=0
