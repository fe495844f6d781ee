"""r06: does the wide-body tracker stay quiet in BASELINE config 5 (and the jack field) now that it runs in worlds with bodies of several components?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
for name, sc in (("config5", scenes.dumbbell_field(64, 16, 64)), ("jacks 32x8x32", scenes.jack_field(32, 8, 32)), ("caterpillars 16x4x16", scenes.caterpillar_field(16, 4, 16))):
    w = mgf_amd.World.from_scene(ctx, sc)
    w.step_many(float(sc["dt"]), sc["iters"], 400)
    print(name, "wide ticks", w.counter("wide_ticks"), "wide bodies", w.counter("wide_bodies"), "overflows", w.counter("wide_overflows"), "front_rows", w.counter("front_rows"))
ctx.close()
