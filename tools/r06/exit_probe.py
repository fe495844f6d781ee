"""r06: does a process that leaves its worlds to the interpreter's finalisation exit cleanly?  (tools/r06/front_rows_check.py aborted inside mgf_world_free at exit)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mgf_amd
from mgf_amd import scenes
mode = sys.argv[1]
ctx = mgf_amd.Context(0)
sc = scenes.capsule_field(128, 32, 32, quads=158) if "big" in mode else scenes.capsule_field(8, 6, 8, quads=12, pitch=1.6)
a, b = mgf_amd.World.from_scene(ctx, sc), mgf_amd.World.from_scene(ctx, sc)
b.set_option("front_rows", 0)
dt, it = float(sc["dt"]), sc["iters"]
sa = a.step_many(dt, it, 60 if "big" in mode else 20); sb = b.step_many(dt, it, 60 if "big" in mode else 20)
if "many" in mode: sa = a.step_many(dt, it, 200)
if "close" in mode: ctx.close()
if "del" in mode: del a, b
print("end of", mode, flush=True)
