"""BASELINE config 4's scene (1 048 576 spheres) as ONE undivided world on one GPU: the exact canonical order (no block-Jacobi seam),
for comparison with the 8-tile figure.  The block-local solver's tables hold at most 2047 bodies per block (256 blocks), so a world
this large is solved by the global dataflow launch (solver mode 1).  Writes gpurun_out/config4_undivided_1gpu.json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mgf_amd
from mgf_amd import scenes

warm, steps = 10, 60
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(128, 128, 64)
w = mgf_amd.World.from_scene(ctx, sc)
w.set_option("phase_timing", 1)  # (ms_solve below)
dt, it = float(sc["dt"]), sc["iters"]
for _ in range(warm):
    w.step(dt, it)
units, ms_solve = 0, 0.0
t0 = time.perf_counter()
for _ in range(steps):
    st = w.step(dt, it)
    units += int(st.n_constraints) * it
    ms_solve += float(st.ms_solve)
el = time.perf_counter() - t0
w2 = mgf_amd.World.from_scene(ctx, sc)
w2.set_option("solver_mode", 0)  # one launch per frontier: an independent executor
for _ in range(warm + steps):
    w2.step(dt, it)
a, b = w.state(), w2.state()
same = all(np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)) for k in ("x", "q", "v", "omega"))
out = {"workload": "BASELINE config 4 scene (1 048 576 spheres, 128x128x64) as one undivided world, canonical order", "n_gpus": 1,
       "warmup": warm, "steps": steps, "ms_per_step": el * 1e3 / steps, "value": units / el, "unit": "constraint-iters/s",
       "ms_solve_per_step": ms_solve / steps, "constraints_last_tick": int(st.n_constraints),
       "block_local_solver_used": bool(w.counter("flow6_max_slots")), "bit_identical_to_launch_per_frontier_after_70_ticks": bool(same)}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/config4_undivided_1gpu.json", "w"))
print(json.dumps(out))
