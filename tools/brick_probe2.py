import sys
sys.path.insert(0, '/root/repo')
import mgf_amd
from mgf_amd import scenes
which = sys.argv[1]
ctx = mgf_amd.Context(0)
sc = scenes.capsule_field(128, 32, 32, quads=158) if which == 'config3' else scenes.dumbbell_field(64, 16, 64)
dt = float(sc['dt'])
for fill in [int(a) for a in sys.argv[2:]] or [0]:
    w = mgf_amd.World.from_scene(ctx, sc)
    w.set_option('phase_timing', 1)
    if fill: w.set_option('cell_fill', fill)
    out = []
    for t in range(40):
        w.step(dt, 10)
        out.append((w.counter('pair_brick_slow_queries'), w.counter('pair_brick_off_ticks'), round(w.stats.ms_broadphase, 3)))
    print(which, 'fill', fill, 'n', len(w), out[:6], out[-3:])
    del w
