#!/bin/bash
# Which of the tracked probe scripts still run against the current library?  Every python probe under tools/scratch and tools/r06 that needs no
# argument, each under `timeout`; status and the last line of output -> gpurun_out/probe_status.txt  (tools/scratch/README.md is written from it).
T=${1:-150}
cd $GRAFT_REPO_ROOT
: > gpurun_out/probe_status.txt
for f in tools/scratch/*.py tools/r06/*.py; do
  case $(basename $f) in gaps.py|timeline.py|cells_dbg.py|dbg_c5.py|dbg_many.py|list_stability.py) echo "$f SKIP needs-arguments-or-a-trace" >> gpurun_out/probe_status.txt; continue;; esac
  s=$(date +%s)
  out=$(timeout $T python $f 2>&1 | tail -1 | cut -c1-160)
  rc=${PIPESTATUS[0]}
  echo "$f rc=$rc $(( $(date +%s) - s ))s | $out" >> gpurun_out/probe_status.txt
done
cat gpurun_out/probe_status.txt
