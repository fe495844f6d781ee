"""The cell grid of a tile-shaped slab (16 x 128 x 64 spheres): which widening of the thin axes / cell fill keeps k_pair_brick running, and what
the broadphase phase costs (HIP events, ticks 20..40)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(16, 128, 64)
dt, it = float(sc["dt"]), sc["iters"]
for frac in (0, 30, 50, 75, 100):
    for fill in (0, 8, 16, 24):
        w = mgf_amd.World.from_scene(ctx, sc)
        if frac: w.set_option("grid_min_frac_pct", frac)
        if fill: w.set_option("cell_fill", fill)
        w.set_option("phase_timing", 1)
        w.step_many(dt, it, 20)
        ms = 0.0; off = 0
        for st in w.step_many(dt, it, 20):
            ms += float(st["ms_broadphase"])
        print(f"min_frac {frac or 'default'} cell_fill {fill or 'default'}: broadphase {ms / 20 * 1e3:.1f} us/tick, levels {int(st['n_levels'])}, brick off for {w.counter('pair_brick_off_ticks')} ticks, slow queries {w.counter('pair_brick_slow_queries')}", flush=True)
        del w
