import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
sc = scenes.capsule_field(8, 6, 8, quads=12, pitch=1.6)
dt, it = float(sc["dt"]), sc["iters"]
for opts in ({}, {"side_stream": 0}, {"front_rows": 0}):
    a = mgf_amd.World.from_scene(ctx, sc)
    for k, v in opts.items(): a.set_option(k, v)
    b = mgf_amd.World.from_scene(ctx, sc)
    for k, v in opts.items(): b.set_option(k, v)
    for batch in range(4):
        many = a.step_many(dt, it, 30)
        singles = [b.step(dt, it).n_constraints for _ in range(30)]
        x, y = a.state(), b.state()
        same = all(np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)) for k in ("x", "q", "v", "omega"))
        print(opts, batch, [int(m["n_constraints"]) for m in many][:4], singles[:4], "...", [int(m["n_constraints"]) for m in many][-2:], singles[-2:], "same state", same)
