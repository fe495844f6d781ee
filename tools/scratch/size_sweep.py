"""sphere piles of sizes the bench never runs (block plans, grid levels and LDS splits change with n): the default tick against the plainest one
(global dataflow solver, list-based front end, no re-sort), bit for bit, 160 ticks each"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
for dims in [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or ((17, 9, 13), (50, 50, 40), (70, 70, 70), (80, 80, 80), (90, 100, 64), (128, 100, 64)):
    sc = scenes.sphere_pile(*dims)
    a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
    for k, v in (("solver_mode", 1), ("fused_contacts", 0), ("resort_every", 0), ("cells_in_integrate", 0)):
        b.set_option(k, v)
    dt, it = float(sc["dt"]), sc["iters"]
    t0 = time.time()
    for s in range(40, 161, 40):
        sa, sb = a.step_many(dt, it, 40), b.step_many(dt, it, 40)
        x, y = a.state(), b.state()
        same = all(np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)) for k in ("x", "q", "v", "omega"))
        assert same and int(sa[39]["n_constraints"]) == int(sb[39]["n_constraints"]), (dims, s)
    print(f"{dims}: {len(a)} bodies, 160 ticks bit-identical; {int(sa[39]['n_constraints'])} constraints at the end; mode 6 ran {a.counter('flow6_runs')} ticks, fell back {a.counter('flow6_fallbacks')}, "
          f"capacity retries {a.counter('capacity_retries')}, re-sorts {a.counter('store_resorts')} [{time.time() - t0:.0f} s]", flush=True)
    del a, b
print("OK")
