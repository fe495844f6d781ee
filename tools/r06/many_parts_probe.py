"""r06: a field of 16-part bodies (scenes.caterpillar_field): ms per tick over the fall and the pile, the solver modes against each other."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd, torch
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
dims = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 8, 32)
sc = scenes.caterpillar_field(*dims)
a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
b.set_option("solver_mode", 1)
dt, it = float(sc["dt"]), sc["iters"]
print(len(a), "bodies of 16 components")
for s in range(50, 601, 50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sa = a.step_many(dt, it, 50)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    sb = b.step_many(dt, it, 50)
    x, y = a.state(), b.state()
    same = all(np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)) for k in ("x", "q", "v", "omega"))
    print(f"tick {s}: {int(sa[49]['n_constraints'])} constraints ({int(sa[49]['n_terrain_constraints'])} terrain), {int(sa[49]['n_pair_candidates'])} partners; {1e3 * (t1 - t0) / 50:.3f} ms/tick; modes 6 / 1 {'bit-identical' if same else 'DIFFER'}; flow6 runs {a.counter('flow6_runs')} fallbacks {a.counter('flow6_fallbacks')}", flush=True)
    assert same
