"""Where k_pair_brick gives up: slow queries (answered from global memory: the cells reach outside the staged box, or the box holds more records
than its LDS copy) per tick on the falling pile, the settled pile and a tile-shaped slab."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
for name, sc, ticks in (("pile 64^3", scenes.sphere_pile(64, 64, 64), (5, 25, 60, 150, 400)), ("slab 16x128x64", scenes.sphere_pile(16, 128, 64), (5, 25, 60, 150))):
    w = mgf_amd.World.from_scene(ctx, sc)
    dt, it = float(sc["dt"]), sc["iters"]
    done = 0
    for t in ticks:
        w.step_many(dt, it, t - done); done = t
        st = w.step(dt, it); done += 1
        print(name, "tick", done, "bodies", len(w), "constraints", int(st.n_constraints), "slow queries", w.counter("pair_brick_slow_queries"), "brick off for", w.counter("pair_brick_off_ticks"), "ticks", flush=True)
    del w
