#!/bin/bash
# development aid: bench.py under a list of library options, one line each (no CPU baseline)
for o in "$@"; do
  args=""; for kv in ${o//,/ }; do args="$args --opt $kv"; done
  r=$(timeout 120 python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1))")
  echo "$o: $r"
done
