"""Development probe: k_pair_grid on config 3 / 5 by cell fill (grid level) with the brick kernel off."""
import sys
sys.path.insert(0, '/root/repo')
import time, torch, mgf_amd
from mgf_amd import scenes
which = sys.argv[1]
ctx = mgf_amd.Context(0)
sc = scenes.capsule_field(128, 32, 32, quads=158) if which == 'config3' else scenes.dumbbell_field(64, 16, 64)
dt = float(sc['dt'])
for fill in [int(a) for a in sys.argv[2:]]:
    w = mgf_amd.World.from_scene(ctx, sc)
    w.set_option('pair_brick', 0); w.set_option('cell_fill', fill); w.set_option('phase_timing', 1)
    w.step_many(dt, 10, 150 if which == 'config3' else 80)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = w.step_many(dt, 10, 60)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 60 * 1e3
    print(which, 'cell_fill', fill, 'ms/tick %.4f' % el, 'broadphase %.4f' % (sum(s.ms_broadphase for s in st) / 60), 'narrow %.4f' % (sum(s.ms_narrowphase for s in st) / 60))
    del w
