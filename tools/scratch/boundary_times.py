"""wall time of the bulk boundary calls on BASELINE config 2 (262 144 spheres)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(64, 64, 64)
dt, it = float(sc["dt"]), sc["iters"]
def T(label, f, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{label}: " + " / ".join(f"{t:.2f}" for t in ts) + " ms", flush=True)
    return r
w = T("World.from_scene (add_bodies + terrain)", lambda: mgf_amd.World.from_scene(ctx, sc), 2)
w.step_many(dt, it, 70)
print("store permuted:", w.counter("store_permuted"))
s = T("state()", w.state)
T("write_state(all)", lambda: w.write_state(x=s["x"], q=s["q"], v=s["v"], omega=s["omega"], delta=s["delta"]))
T("colliders()", w.colliders)
T("tags()", w.tags)
T("set_tags()", lambda: w.set_tags(np.arange(len(w), dtype=np.uint32)))
w.step(dt, it)
T("constraints()", w.constraints, 2)
T("clone()", w.clone, 2)
T("step()", lambda: w.step(dt, it), 5)
