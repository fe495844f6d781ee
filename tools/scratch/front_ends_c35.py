"""BASELINE configs 3 and 5 over 400 ticks: the default front end against the exact ones (two-pass candidate lists, tree walks instead of the cell
grids, the global solver), bit for bit"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
for name, sc in (("config3", scenes.capsule_field(128, 32, 32, quads=158)), ("config5", scenes.dumbbell_field(64, 16, 64))):
    for alt in ({"two_pass_candidates": 1, "solver_mode": 1}, {"broadphase_tree": 1, "terrain_tree": 1, "resort_every": 0}):
        a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
        for k, v in alt.items(): b.set_option(k, v)
        dt, it = float(sc["dt"]), sc["iters"]
        t0 = time.time()
        for s in range(100, 401, 100):
            sa, sb = a.step_many(dt, it, 100), b.step_many(dt, it, 100)
            x, y = a.state(), b.state()
            same = all(np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)) for k in ("x", "q", "v", "omega"))
            assert same and int(sa[99]["n_constraints"]) == int(sb[99]["n_constraints"]), (name, alt, s)
        print(f"{name} against {alt}: 400 ticks bit-identical, {int(sa[99]['n_constraints'])} constraints at the end [{time.time() - t0:.0f} s]", flush=True)
        del a, b
print("OK")
