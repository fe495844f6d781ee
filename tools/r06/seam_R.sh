#!/bin/bash
# seam depth across tile faces against R (ghost velocity refresh period): bash tools/r06/seam_R.sh "<R values>" <warmup ticks> [scene]
for R in $1; do python bench.py --scene ${3:-config4} --refresh-every $R --no-cpu-baseline --no-settled-tiles --warmup $2 --steps 60 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d['seam_penetration']
print('R=$R warmup $2: ms/step %.3f; across faces mean %.4f p99 %.4f max %.3f (%d pairs); inside mean %.4f p99 %.4f' % (d['ms_per_step'], s['across_tile_faces']['mean'], s['across_tile_faces']['p99'], s['across_tile_faces']['max'], s['across_tile_faces']['pairs'], s['inside_tiles']['mean'], s['inside_tiles']['p99']))"
done
