"""four times BASELINE configs 3 and 5 (524 288 capsules over a heightfield; 262 144 two-part bodies), 200 ticks: default against the exact front end"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
for name, sc in (("capsules x4", scenes.capsule_field(256, 32, 64, quads=316)), ("two-part bodies x4", scenes.dumbbell_field(128, 16, 128))):
    a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
    for k, v in {"two_pass_candidates": 1, "solver_mode": 1, "resort_every": 0}.items(): b.set_option(k, v)
    dt, it = float(sc["dt"]), sc["iters"]
    t0 = time.time()
    for s in range(50, 201, 50):
        sa, sb = a.step_many(dt, it, 50), b.step_many(dt, it, 50)
        x, y = a.state(), b.state()
        same = all(np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)) for k in ("x", "q", "v", "omega"))
        assert same and int(sa[49]["n_constraints"]) == int(sb[49]["n_constraints"]), (name, s)
    print(f"{name}: {len(a)} bodies, 200 ticks bit-identical, {int(sa[49]['n_constraints'])} constraints at the end; retries {a.counter('capacity_retries')}, mode 6 ran {a.counter('flow6_runs')} [{time.time() - t0:.0f} s]", flush=True)
    del a, b
print("OK")
