"""Development probe: tick time against the cell grid's fill threshold for worlds of several sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
cases = {"pile 64^3": lambda: scenes.sphere_pile(64, 64, 64), "pile 66x64x64": lambda: scenes.sphere_pile(66, 64, 64),
         "pile 80^3": lambda: scenes.sphere_pile(80, 80, 80), "pile 50^3": lambda: scenes.sphere_pile(50, 50, 50),
         "pile 40^3": lambda: scenes.sphere_pile(40, 40, 40), "config 3 (capsules)": lambda: scenes.config(2)}
for name, make in cases.items():
    sc = make()
    if sc is None: continue
    for fill in (8, 16, 32):
        w = mgf_amd.World.from_scene(ctx, sc)
        w.set_option('phase_timing', 1)
        w.set_option("cell_fill", fill)
        bp, tot = [], []
        for s in range(40):
            st = w.step(float(sc["dt"]), 10)
            if s >= 10: bp.append(st.ms_broadphase); tot.append(st.ms_total)
        print(f"{name}: n={st.n_bodies} cell_fill={fill}/8: broadphase {np.mean(bp):.3f} ms, tick {np.mean(tot):.3f} ms", flush=True)
