"""Config 4 undivided (1 048 576 spheres as one world) under the global executors: solver mode 1 (pinned lanes) against 4 (four slots per lane)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(128, 128, 64)
dt, it = float(sc["dt"]), sc["iters"]
res = {}
states = {}
for mode in (1, 4):
    w = mgf_amd.World.from_scene(ctx, sc)
    w.set_option("solver_mode", mode)
    w.set_option("phase_timing", 1)
    for _ in range(10):
        w.step(dt, it)
    ms_solve = 0.0
    t0 = time.perf_counter()
    for _ in range(20):
        st = w.step(dt, it)
        ms_solve += float(st.ms_solve)
    el = time.perf_counter() - t0
    res[mode] = {"ms_per_step": el * 1e3 / 20, "ms_solve": ms_solve / 20, "constraints": int(st.n_constraints)}
    states[mode] = w.state()
    print(mode, res[mode], flush=True)
    del w
a, b = states[1], states[4]
print("bit-identical", all(np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)) for k in ("x", "q", "v", "omega")))
