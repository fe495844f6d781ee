"""iteration counts the bench never uses (0, 1, odd, the block-local solver's limit of 64 and beyond) and a changing dt: default against the global solver"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(40, 40, 40)
for iters in (0, 1, 3, 7, 63, 64, 65, 100):
    a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
    b.set_option("solver_mode", 1); b.set_option("fused_contacts", 0)
    for t in range(30):
        dt = float(sc["dt"]) * (1.0 + 0.3 * ((t % 5) - 2) / 2.0)
        if t % 7 == 3:
            sa, sb = a.step_many(dt, iters, 3)[-1], b.step_many(dt, iters, 3)[-1]
            ca, cb = int(sa["n_constraints"]), int(sb["n_constraints"])
        else:
            sa, sb = a.step(dt, iters), b.step(dt, iters)
            ca, cb = int(sa.n_constraints), int(sb.n_constraints)
        assert ca == cb, (iters, t)
    x, y = a.state(), b.state()
    same = all(np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)) for k in ("x", "q", "v", "omega"))
    print(f"iters {iters}: {ca} constraints, bit-identical {same}; mode 6 ran {a.counter('flow6_runs')}, fell back {a.counter('flow6_fallbacks')}", flush=True)
    assert same
print("OK")
