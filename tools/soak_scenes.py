"""Long-run cross-check of the default solver (mode 6: quad trips, message channels) against the global dataflow launch (mode 1) on the
BASELINE scenes that are not spheres: capsules over the heightfield (config 3) and two-part bodies (config 5); bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mgf_amd
from mgf_amd import scenes
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 600
every = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = mgf_amd.Context(0)
for name, sc in (("config3", scenes.capsule_field(128, 32, 32, quads=158)), ("config5", scenes.dumbbell_field(64, 16, 64)),
                 ("8192 bodies of 16 components", scenes.caterpillar_field(32, 8, 32))):  # (r06: bodies whose parts live in the pool)
    a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
    a.set_option("solver_mode", 1)
    a.set_option("front_rows", 0); a.set_option("wide_list", 0); a.set_option("side_stream", 0)  # (r06: the list-based front end on one stream too - the plainest tick against the default one)
    dt, it = float(sc["dt"]), sc["iters"]
    t0 = time.time()
    for s in range(every, ticks + 1, every):
        sa, sb = a.step_many(dt, it, every), b.step_many(dt, it, every)
        assert int(sa[every - 1]["n_constraints"]) == int(sb[every - 1]["n_constraints"])
        x, y = a.state(), b.state()
        for k in ("x", "q", "v", "omega"):
            assert np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)), f"{name} tick {s}: {k} differs"
        print(f"{name} tick {s}: {int(sb[every - 1]['n_constraints'])} constraints, modes 1 / 6 bit-identical; mode 6 ran {b.counter('flow6_runs')} ticks, fell back {b.counter('flow6_fallbacks')} [{time.time() - t0:.0f} s]", flush=True)
print("soak OK")
