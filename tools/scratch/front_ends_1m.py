"""the million-sphere pile undivided, 300 ticks through its collapse: the fused front end (k_contacts_spheres) against the list-based one, bit for bit"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(128, 128, 64)
a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
b.set_option("fused_contacts", 0)
dt, it = float(sc["dt"]), sc["iters"]
for s in range(50, 301, 50):
    sa, sb = a.step_many(dt, it, 50), b.step_many(dt, it, 50)
    x, y = a.state(), b.state()
    same = all(np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)) for k in ("x", "q", "v", "omega"))
    print(f"tick {s}: {int(sa[49]['n_constraints'])} / {int(sb[49]['n_constraints'])} constraints, states bit-identical: {same}", flush=True)
    assert same
print("OK")
