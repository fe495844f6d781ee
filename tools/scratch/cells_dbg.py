import sys, numpy as np
sys.path.insert(0, '.')
import mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
for name, sc in (("balls", scenes.balls_demo(8)), ("pile", scenes.sphere_pile(16, 8, 16))):
    dt, it = float(sc["dt"]), sc["iters"]
    a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
    a.set_option("cells_in_integrate", 0)
    for k in range(30):
        sa, sb = a.step(dt, it), b.step(dt, it)
        x, y = a.state(), b.state()
        same = all(np.array_equal(x[f].view(np.uint32), y[f].view(np.uint32)) for f in ("x", "q", "v", "omega"))
        print(name, k, int(sa.n_constraints), int(sb.n_constraints), int(sa.n_pair_candidates), int(sb.n_pair_candidates), int(sa.n_terrain_constraints), int(sb.n_terrain_constraints), same, b.counter("capacity_retries"), b.counter("grid_too_wide"), flush=True)
        if not same: break
