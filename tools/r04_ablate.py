"""Timing of the solver launch under a differently built library (tools/build_variant.sh), on the state another process saved:
    python tools/r04_ablate.py save            # normal library: warm the scenes up, save the states
    MGF_AMD_LIB=mgf_amd/variants/libmgf_hip_X.so python tools/r04_ablate.py time   # the variant on the same states
    python tools/r04_ablate.py time            # ... and the normal library, for reference
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import mgf_amd  # noqa: E402
from mgf_amd import scenes  # noqa: E402

SC = {"config2": (lambda: scenes.sphere_pile(64, 64, 64), 15), "config2_settled": (lambda: scenes.sphere_pile(64, 64, 64), 410),
      "config3": (lambda: scenes.capsule_field(128, 32, 32, quads=158), 160), "config5": (lambda: scenes.dumbbell_field(64, 16, 64), 90)}
mode = sys.argv[1]
names = sys.argv[2].split(",") if len(sys.argv) > 2 else list(SC)
opts = [kv.split("=") for kv in sys.argv[3].split(",")] if len(sys.argv) > 3 else []
ctx = mgf_amd.Context(0)
for name in names:
    build, warm = SC[name]
    sc = build()
    dt, iters = float(sc["dt"]), 10
    w = mgf_amd.World.from_scene(ctx, sc)
    for k, v in opts:
        w.set_option(k, int(v))
    path = f"/tmp/r04_state_{name}.npz"
    if mode == "save":
        w.step_many(dt, iters, warm)
        st = w.state()
        np.savez(path, **{k: st[k] for k in ("x", "q", "v", "omega")})
        print("saved", name, path)
        continue
    st = np.load(path)
    w.write_state(x=st["x"], q=st["q"], v=st["v"], omega=st["omega"])
    w.step_many(dt, iters, 3)
    w.set_option("time_solver_kernels", 1)
    per = w.step_many(dt, iters, 10)
    cons = np.mean([int(p["n_constraints"]) for p in per])
    us = np.mean([float(p["ms_solver_kernels"]) for p in per]) * 1e3
    print(f"{os.environ.get('MGF_AMD_LIB', 'default'):50s} {name:16s} constraints {cons:9.0f}  solver {us:7.1f} us/tick  RL{w.counter('flow6_rec_lds')} CL{w.counter('flow6_const_lds')}", flush=True)
