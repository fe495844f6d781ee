"""Tick timing of BASELINE config 3 (131 072 capsules over a 49 928-triangle heightfield) - development aid."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
sc = scenes.capsule_field(128, 32, 32, quads=158)
w = mgf_amd.World.from_scene(ctx, sc)
w.set_option('phase_timing', 1)
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 80
t = []; ph = None
for s in range(ticks):
    t0 = time.perf_counter(); st = w.step(float(sc['dt']), 10); t.append(time.perf_counter() - t0)
    if s == ticks - 1: ph = st.as_dict()
print(f"config 3: {len(w)} capsules; last 40 ticks mean {np.mean(t[-40:])*1e3:.3f} ms/tick; constraints {ph['n_constraints']} (terrain {ph['n_terrain_constraints']})")
print({k: round(v, 3) for k, v in ph.items() if k.startswith('ms_')})
