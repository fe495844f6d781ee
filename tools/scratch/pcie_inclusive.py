"""The tick as a host-resident caller sees it (the reference keeps its RigidBodyVec in host memory): World::step per call, and after every
tick the bodies' x, q, v, omega read back to host arrays (mgf_world_read_state: 52 bytes per body over PCIe) - against the resident tick.
Never bench.py's `value`: a figure for DESIGN.md section 7."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(64, 64, 64)
dt, it = float(sc["dt"]), sc["iters"]
w = mgf_amd.World.from_scene(ctx, sc)
w.step_many(dt, it, 5)
snap = w.clone()
for mode in ("resident", "read back every tick", "read back and written every tick", "resident", "read back every tick", "read back and written every tick"):
    x = snap.clone()
    cons, nbytes = 0, 0
    t0 = time.perf_counter()
    for _ in range(20):
        st = x.step(dt, it)
        cons += int(st.n_constraints)
        if mode != "resident":
            s = x.state()
            nbytes += sum(s[k].nbytes for k in ("x", "q", "v", "omega", "delta"))
            if "written" in mode:
                x.write_state(x=s["x"], q=s["q"], v=s["v"], omega=s["omega"])
                nbytes += sum(s[k].nbytes for k in ("x", "q", "v", "omega"))
    ms = (time.perf_counter() - t0) * 1e3 / 20
    print(f"{mode}: {ms:.4f} ms per tick, {cons * it / (ms * 20e-3) / 1e9:.2f} G constraint-iters/s, {nbytes / 20 / 1e6:.1f} MB over PCIe per tick", flush=True)
    del x
