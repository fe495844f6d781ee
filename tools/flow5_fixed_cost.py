"""Fixed vs per-iteration cost of the block-local solver launch (development aid): solve phase time for 1, 2, 5, 10 iterations."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, mgf_amd
from mgf_amd import scenes
ctx = mgf_amd.Context(0)
sc = scenes.sphere_pile(64, 64, 64)
for mode in (5, 1):
    for iters in (1, 2, 5, 10, 20):
        w = mgf_amd.World.from_scene(ctx, sc)
        w.set_option('phase_timing', 1)
        w.set_option('solver_mode', mode)
        w.set_option('time_solver_kernels', 1)
        ms = []; km = []
        for s in range(30):
            st = w.step(float(sc['dt']), iters)
            if s >= 10: ms.append(st.ms_solve); km.append(st.ms_solver_kernels)
        print(f"mode {mode} iters {iters:2d}: solve phase {np.mean(ms):.3f} ms, solver kernels {np.mean(km):.3f} ms", flush=True)
        del w
