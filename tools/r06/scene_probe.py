"""r06: what the cell grid of a scene is laid over: extents, largest fat half extents, levels (tools/r06/scene_probe.py <config3|config5|config2> <ticks>)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
name, ticks = sys.argv[1], int(sys.argv[2])
sc = {"config3": lambda: scenes.capsule_field(128, 32, 32, quads=158), "config5": lambda: scenes.dumbbell_field(64, 16, 64), "config2": lambda: scenes.sphere_pile(64, 64, 64)}[name]()
w = mgf_amd.World.from_scene(ctx, sc)
dt, it = float(sc["dt"]), sc["iters"]
done = 0
for t in range(0, ticks, 50):
    w.step_many(dt, it, 50); done += 50
    print(name, "tick", done, "ext", [w.counter(f"scene_ext_milli_{a}") / 1000 for a in "xyz"], "rmax", [w.counter(f"scene_rmax_milli_{a}") / 1000 for a in "xyz"], "levels", w.counter("grid_levels"), flush=True)
