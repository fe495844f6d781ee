"""what the cubic grid rule chooses, tick by tick (option grid_cubic): levels, cell edge, permutation, the brick kernel's slow queries"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
for name, sc, marks in (("pile", scenes.sphere_pile(64, 64, 64), (3, 5, 10, 25, 60, 100, 200, 400, 420)), ("slab", scenes.sphere_pile(16, 128, 64), (3, 10, 30, 70, 130))):
    for cubic, fill in ((0, 7), (1, 7), (1, 9)):
        w = mgf_amd.World.from_scene(ctx, sc)
        w.set_option("grid_cubic", cubic); w.set_option("grid_cubic_fill", fill); w.set_option("phase_timing", 1)
        dt, it = float(sc["dt"]), sc["iters"]
        done = 0
        for m in marks:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for st in w.step_many(dt, it, m - done): pass
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / (m - done)
            done = m
            print(f"{name} cubic={cubic} fill={fill} tick {m}: {ms:.3f} ms/tick, levels {w.counter('grid_levels')}, cell {w.counter('grid_cell_micro') / 1e6:.3f}, perm {w.counter('grid_perm'):#x}, "
                  f"slow queries {w.counter('pair_brick_slow_queries')}, brick off for {w.counter('pair_brick_off_ticks')}, constraints {int(st['n_constraints'])}, broadphase {float(st['ms_broadphase']) * 1e3:.0f} us", flush=True)
        del w
