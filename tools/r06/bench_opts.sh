#!/bin/bash
# wall-clock bench lines of one window under different options: bash tools/r06/bench_opts.sh "<none|opt=a,opt=b> ..." <bench args...>
VS=$1; shift
for V in $VS; do
  OPTS=""; if [ $V != none ]; then for o in ${V//,/ }; do OPTS="$OPTS --opt $o"; done; fi
  python bench.py --no-cpu-baseline --no-settled --no-order-check --no-other-configs "$@" $OPTS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V', 'ms/step %.4f' % d['ms_per_step'], 'value %.3e' % d['value'], 'solver us', d['roofline'].get('avg_launch_us'))"
done
