"""Development probe: does k_pair_brick hold on a scene (queries answered from global memory per tick, back-off), by cell fill."""
import sys
sys.path.insert(0, '/root/repo')
import time, torch
import mgf_amd
from mgf_amd import scenes
which = sys.argv[1] if len(sys.argv) > 1 else 'config3'
ctx = mgf_amd.Context(0)
sc = scenes.capsule_field(128, 32, 32, quads=158) if which == 'config3' else scenes.dumbbell_field(64, 16, 64)
dt = float(sc['dt'])
for fill in [int(a) for a in sys.argv[2:]] or [0]:
    w = mgf_amd.World.from_scene(ctx, sc)
    if fill: w.set_option('cell_fill', fill)
    w.step_many(dt, 10, 150 if which == 'config3' else 80)
    slow = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60):
        w.step(dt, 10); slow.append(w.counter('pair_brick_slow_queries'))
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 60 * 1e3
    print(which, 'cell_fill', fill, 'ms/tick %.4f' % el, 'slow queries per tick (last 5):', slow[-5:], 'brick off ticks', w.counter('pair_brick_off_ticks'), 'ms_broadphase %.4f' % w.stats.ms_broadphase)
    del w
